"""Round 5: where one wave per source point stops paying against one lane per point in the device-resident optimiser launch
(sp_gicp_align_optimize, Gauss-Newton, 10 iterations, criteria 0), on the benchmark's synthetic clouds at config-4 density and
on clouds of surfaces (sp_gicp_source_set_wave_per_point 2 against 0). GPU box: python scratch/waveq_crossover.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import gicp_pair

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
ITERS = 10

def timed(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms)), out

for surfaces in (False, True):
    for n in (3000, 8000, 20000, 50000, 100000, 250000):
        r = 10.0 * (n / 1e6) ** (1.0 / 3.0)
        src, tgt, T_gt = gicp_pair(n, r, 7)
        if surfaces:  # three sheets: tens of points per occupied cell at the volume rule's cell size
            for c in (src, tgt):
                c[:, 2] = np.round(c[:, 2] / r * 1.5) * r / 1.5 + 0.002 * c[:, 2]
        S_pts, T_pts = dev(src), dev(tgt)
        S, Tg = sp.PointCloudShared(S_pts), sp.PointCloudShared(T_pts)
        for c in (S, Tg):
            c.covs = sp.GridKNN.build(c.points, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
        grid = sp.GridKNN.build(Tg.points, points_per_cell=0.5)
        prep = sp.PreparedTarget(grid, Tg.covs)
        row = []
        for wq in (2, 0):
            reg = sp.Registration(sp.RegistrationParams(max_iterations=ITERS, criteria_rotation=0.0, criteria_translation=0.0,
                                                        optimization_method="GN"))
            reg._set_source_option("opt_wave_query", wq)
            ms, res = timed(lambda: reg.align_optimize(S, prep, None, None, True))
            row.append((ms / ITERS * 1e3, res.searched))
        print(f"{'surfaces' if surfaces else 'filled box'} n {n:7d} fullest cell {grid.max_cell_points():4d}: wave per point {row[0][0]:7.1f} us / iteration, "
              f"lane per point {row[1][0]:7.1f}  (searched {row[0][1]} / {row[1][1]})")
