"""sp_bvh_self_knn: count / collect / rank (bvh_self_heap_kernel) against the sorted-insertion kernel — same lists, time."""
import os
import sys

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


raw = open(os.path.join(ROOT, "tests", "golden", "target.ply"), "rb").read()
head, body = raw.split(b"end_header\n", 1)
n = int([l for l in head.split(b"\n") if l.startswith(b"element vertex")][0].split()[-1])
scan = np.ones((n, 4), np.float32)
scan[:, :3] = np.frombuffer(body, dtype="<f4", count=n * 4).reshape(n, 4)[:, :3]
from sycl_points_amd.synthetic import Mt19937Cloud  # noqa: E402

dup = nonuniform_cloud(200_000)
dup[1000:1500] = dup[1000]  # 500 copies of one point
dup[5000:5030] = dup[5000]
clouds = (("raw scan 69088", scan), ("non-uniform 1M", nonuniform_cloud(1_000_000)),
          ("uniform 1M", Mt19937Cloud(1234).uniform_points(1_000_000, 10.0)), ("duplicates 200k", dup),
          ("small 300", nonuniform_cloud(300)))
ks = [int(a) for a in sys.argv[1:]] or [20, 10, 5, 2]
for name, pts in clouds:
    P = torch.from_numpy(pts).cuda()
    b = sp.BVH.build(P)
    line = [name + ":"]
    for k in ks:
        b._set_option("bvh_self_heap", 0)
        ref = b.self_knn(k)
        t_old = med(lambda: b.self_knn(k))
        b._set_option("bvh_self_heap", 1)
        new = b.self_knn(k)
        t_new = med(lambda: b.self_knn(k))
        same = torch.equal(ref.indices, new.indices) and torch.equal(ref.distances, new.distances)
        bad = int((ref.indices != new.indices).any(1).sum())
        rq = sp.KNNResult()
        b._set_option("bvh_self_heap", 0)
        b.knn_search_async(P, k, rq)
        qi0, qd0 = rq.indices.clone(), rq.distances.clone()
        tq_old = med(lambda: b.knn_search_async(P, k, rq))
        rr0 = b.radius_search(P, k, 0.3)
        tr_old = med(lambda: b.radius_search_async(P, k, 0.3, rq))
        b._set_option("bvh_self_heap", 1)
        b.knn_search_async(P, k, rq)
        sameq = torch.equal(qi0, rq.indices) and torch.equal(qd0, rq.distances) and torch.equal(qi0, ref.indices)
        tq_new = med(lambda: b.knn_search_async(P, k, rq))
        b._set_option("bvh_sort_queries", 0)
        b.knn_search_async(P, k, rq)
        sameq = sameq and torch.equal(qi0, rq.indices) and torch.equal(qd0, rq.distances)
        tq_unsorted = med(lambda: b.knn_search_async(P, k, rq))
        b._set_option("bvh_sort_queries", 1)
        rr1 = b.radius_search(P, k, 0.3)
        samer = torch.equal(rr0.indices, rr1.indices) and torch.equal(rr0.distances, rr1.distances)
        tr_new = med(lambda: b.radius_search_async(P, k, 0.3, rq))
        line.append(f"\n   k={k}: self {t_old:.3f} -> {t_new:.3f} ms identical {same} ({bad} rows differ); query-order {tq_old:.3f} -> heap {tq_unsorted:.3f} -> heap + sorted queries {tq_new:.3f} ms identical {sameq}; radius 0.3 {tr_old:.3f} -> {tr_new:.3f} ms identical {samer}")
    print("".join(line), flush=True)
