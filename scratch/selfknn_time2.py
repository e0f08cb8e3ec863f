"""self-kNN on 1 M uniform points: median ms per k and output set (lists / covariances), to-do share of the select kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud

def med(fn, runs=11):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(runs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))

n = 1_000_000
P = torch.from_numpy(Mt19937Cloud(1234).uniform_points(n, 10.0)).cuda()
for ppc in [float(a) for a in sys.argv[1:]] or [6.0]:
    grid = sp.GridKNN.build(P, points_per_cell=ppc)
    for k in (7, 8, 10):
        for mode in (0, 3):
            grid._set_option("self_knn_mode", mode) if hasattr(grid, "_set_option") else None
            t1 = med(lambda: grid.self_knn(k, True, False, False))
            t2 = med(lambda: grid.self_knn(k, False, True, False))
            print(f"ppc {ppc} k {k} mode {mode}: lists {t1:.3f} ms, covariances {t2:.3f} ms", flush=True)
