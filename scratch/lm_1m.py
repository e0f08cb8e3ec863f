"""LM + Geman-McClure at 1 M points through sp_gicp_align_optimize (bench.py's lm_geman_mcclure block alone), for A/B runs of two
libraries on one box: SP_LIB=<path to libsycl_points_amd.so> python scratch/lm_1m.py"""
import os, sys, shutil
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("SP_LIB"):
    shutil.copy(os.environ["SP_LIB"], os.path.join(ROOT, "sycl_points_amd", "lib", "libsycl_points_amd.so"))
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import gicp_pair

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
n = 1_000_000
src, tgt, T_gt = gicp_pair(n, 10.0)
Tg = sp.PointCloudShared(dev(tgt))
Tg.covs = sp.GridKNN.build(Tg.points, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
grid = sp.GridKNN.build(Tg.points, points_per_cell=0.5)
S_all = dev(src)
S_all = S_all[sp.GridKNN.build(S_all, points_per_cell=1.0).order()].contiguous()
covs = sp.GridKNN.build(S_all, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
S = sp.PointCloudShared(S_all, covs=covs)
prep = sp.PreparedTarget(grid, Tg.covs)
for label, scales in (("one level", [10.0]), ("three levels", [10.0, 5.0, 2.5])):
    reg = sp.Registration(sp.RegistrationParams(robust_type="GEMAN_MCCLURE", optimization_method="LM", max_iterations=10))
    ms = []
    for _ in range(11):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); res = reg.align_optimize(S, prep, None, scales, "presorted"); e1.record()
        torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    print(f"{label}: {np.median(ms[2:]):.4f} ms  lin {res.linearizations} trials {res.trials} searched {res.searched}")
