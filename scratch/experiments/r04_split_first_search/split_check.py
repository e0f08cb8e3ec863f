"""split_first = 1 (searches of the first linearisation as a launch of their own) against 0: same bits?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import gicp_pair
import test_gpu_persistent_tail as tp

for n in (60_000, 300_000, 1_000_000):
    src, tgt, T_gt = gicp_pair(n, 10.0 * (n / 1e6) ** (1.0 / 3.0))
    dev = tp.dev
    Tg = sp.PointCloudShared(dev(tgt))
    Tg.covs = sp.GridKNN.build(Tg.points, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
    S_all = dev(src)
    S_all = S_all[sp.GridKNN.build(S_all, points_per_cell=1.0).order()].contiguous()
    covs = sp.GridKNN.build(S_all, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
    S = sp.PointCloudShared(S_all, covs=covs)
    prep = sp.PreparedTarget(sp.GridKNN.build(Tg.points, points_per_cell=0.5), Tg.covs)
    for crit, it in ((0.0, 6), (1e-3, 20)):
        a = tp.run(sp, S, prep, crit, it, {"split_first": 0})
        b = tp.run(sp, S, prep, crit, it, {"split_first": 1})
        ok = all(np.array_equal(a[k], b[k]) for k in ("T", "lin", "delta", "idx", "d2")) and a["iters"] == b["iters"]
        print(n, crit, it, "identical" if ok else "DIFFERENT", a["iters"], np.abs(a["T"] - b["T"]).max(), flush=True)
