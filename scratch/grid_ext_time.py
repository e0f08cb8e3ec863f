"""GridKNN with EXTERNAL queries (knn_search, k = 1 / 10 / 20) against the self-kNN of the same cloud and the BVH."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud
def med(fn, runs=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(runs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
g = Mt19937Cloud(1234)
T = torch.from_numpy(g.uniform_points(1_000_000, 10.0)).cuda()
Q = torch.from_numpy(g.uniform_points(1_000_000, 10.0)).cuda()
Qs = Q[sp.GridKNN.build(Q, points_per_cell=1.0).order()].contiguous()
grid = sp.GridKNN.build(T, points_per_cell=6.0)
bvh = sp.BVH.build(T)
r = sp.KNNResult()
for k in (1, 10, 20):
    print(f"k={k}: grid external random-order {med(lambda: grid.knn_search(Q, k)):.3f} ms, cell-ordered queries {med(lambda: grid.knn_search(Qs, k)):.3f} ms; "
          f"bvh external {med(lambda: bvh.knn_search_async(Q, k, r)):.3f} / ordered {med(lambda: bvh.knn_search_async(Qs, k, r)):.3f} ms; "
          f"grid self {med(lambda: grid.self_knn(k, True, False, False)) if k > 1 else float('nan'):.3f} ms", flush=True)
