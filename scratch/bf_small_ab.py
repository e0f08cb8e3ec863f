"""knn_bf_small_kernel alone: the example's downsampled target (6096 voxel means of a scan) and uniform clouds, k = 1 / 10 / 20, for
A/B runs of two libraries on one box (scratch/ab_py.sh)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud

def med(fn, n=31):
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return 1e3 * float(np.median(ms[5:]))
def read(path):
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    n = int([l for l in head.split(b"\n") if l.startswith(b"element vertex")][0].split()[-1])
    a = np.frombuffer(body, dtype="<f4", count=n * 4).reshape(n, 4)
    p = np.ones((n, 4), np.float32); p[:, :3] = a[:, :3]
    return p
pts = read(os.path.join(ROOT, "tests/golden/target.ply"))
linf = np.abs(pts[:, :3]).max(1)
scan = sp.VoxelGrid(0.25).downsampling(torch.from_numpy(pts[(linf >= 0.5) & (linf <= 50)]).cuda()).points.contiguous()
out = []
for name, P in (("scan6096", scan), ("uniform6000", torch.from_numpy(Mt19937Cloud(7).uniform_points(6000, 10.0)).cuda()),
                ("uniform2000", torch.from_numpy(Mt19937Cloud(8).uniform_points(2000, 10.0)).cuda())):
    out.append(name + " " + " ".join(f"k{k} {med(lambda: sp.knn_search_bruteforce(P, P, k)):.1f}" for k in (1, 10, 20)))
print(" | ".join(out), "us")
