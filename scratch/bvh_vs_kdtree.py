"""BVH (device build) vs KD-tree (host build) vs GridKNN on the clouds the uniform grid is bad at: the bundled raw LiDAR scan
(tests/golden/target.ply) and 1M points on planes + a dense cluster. Build and self-kNN times, median of 7, HIP events."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sycl_points_amd.api as sp  # noqa: E402
from test_gpu_bvh import nonuniform_cloud  # noqa: E402


def med(fn, runs=7):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(runs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1))
    return float(np.median(t))


def wall(fn, runs=5):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(runs):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(t))


raw = open(os.path.join(ROOT, "tests", "golden", "target.ply"), "rb").read()
head, body = raw.split(b"end_header\n", 1)
n = int([l for l in head.split(b"\n") if l.startswith(b"element vertex")][0].split()[-1])
scan = np.ones((n, 4), np.float32)
scan[:, :3] = np.frombuffer(body, dtype="<f4", count=n * 4).reshape(n, 4)[:, :3]
for name, pts in (("raw scan 69088", scan), ("non-uniform 1M", nonuniform_cloud(1_000_000)), ("uniform 1M", None)):
    if pts is None:
        from sycl_points_amd.synthetic import Mt19937Cloud
        pts = Mt19937Cloud(1234).uniform_points(1_000_000, 10.0)
    P = torch.from_numpy(pts).cuda()
    t_bvh_build = wall(lambda: sp.BVH.build(P))
    t_kd_build = wall(lambda: sp.KDTree.build(pts), 3)
    t_grid_build = wall(lambda: sp.GridKNN.build(P, points_per_cell=6.0))
    b, kd, g = sp.BVH.build(P), sp.KDTree.build(pts), sp.GridKNN.build(P, points_per_cell=6.0)
    line = [f"{name}: build ms  bvh {t_bvh_build:.3f}  kdtree(host) {t_kd_build:.2f}  grid {t_grid_build:.3f} |"]
    r = sp.KNNResult()
    for k in (1, 10, 20):
        tb = med(lambda: b.self_knn(k))
        tq = med(lambda: b.knn_search_async(P, k, r))
        tk = med(lambda: kd.knn_search_async(P, k, r), 3)
        tg = med(lambda: g.self_knn(k, True, False, False), 3) if (k > 1 and pts.shape[0] <= 1_000_000) else float("nan")
        line.append(f" k={k}: bvh self {tb:.3f} / query-order {tq:.3f}  kdtree {tk:.3f}  grid self {tg:.3f} |")
    print("".join(line), flush=True)
