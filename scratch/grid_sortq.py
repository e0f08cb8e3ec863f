"""GridKNN external queries: as given against sorted by cell (same lists), 1 M queries in random order."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud


def med(fn, runs=7):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(runs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1))
    return float(np.median(t))


pts = Mt19937Cloud(1234).uniform_points(1_000_000, 10.0)
qs = Mt19937Cloud(99).uniform_points(1_000_000, 10.2)
qs[5, 0] = np.nan
P, Q = torch.from_numpy(pts).cuda(), torch.from_numpy(qs).cuda()
Qc = Q[sp.GridKNN.build(Q, points_per_cell=1.0).order()].contiguous()
for ppc, ks in ((6.0, (20, 10)), (2.0, (1, 5))):
    g = sp.GridKNN.build(P, points_per_cell=ppc)
    for k in ks:
        r = sp.KNNResult()
        g._set_option("grid_sort_queries", 0)
        g.knn_search_async(Q, k, r); a_i, a_d = r.indices.clone(), r.distances.clone()
        t0 = med(lambda: g.knn_search_async(Q, k, r))
        tc = med(lambda: g.knn_search_async(Qc, k, r))
        g._set_option("grid_sort_queries", 1)
        g.knn_search_async(Q, k, r)
        same = torch.equal(a_i, r.indices) and torch.equal(a_d, r.distances)
        t1 = med(lambda: g.knn_search_async(Q, k, r))
        rr0 = g.radius_search(Q, k, 0.5) if hasattr(g, "radius_search") else None
        print(f"ppc {ppc} k={k}: random order {t0:.3f} ms -> sorted by cell {t1:.3f} ms (same queries stored in cell order: {tc:.3f}) identical {same}", flush=True)
