"""per-point search time distribution inside gicp_align_kernel, launch by launch (dev build with -DSP_SEARCH_DBG)"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import gicp_pair
import test_gpu_persistent_tail as tp
n = 1_000_000
src, tgt, T_gt = gicp_pair(n, 10.0)
dev = tp.dev
Tg = sp.PointCloudShared(dev(tgt))
Tg.covs = sp.GridKNN.build(Tg.points, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
S_all = dev(src)
S_all = S_all[sp.GridKNN.build(S_all, points_per_cell=1.0).order()].contiguous()
covs = sp.GridKNN.build(S_all, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
S = sp.PointCloudShared(S_all, covs=covs)
prep = sp.PreparedTarget(sp.GridKNN.build(Tg.points, points_per_cell=0.5), Tg.covs)
L = sp._lib.lib()
L.sp_internal_dbg_read.restype = C.c_int
buf = (C.c_uint * 16)()
for iters in (1, 2, 3):
    L.sp_internal_dbg_read(None, 1)
    p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=iters)
    reg = sp.Registration(p)
    reg._set_source_option("persistent", 0)
    reg.align_fused_loop(S, prep, sort_by_cell="presorted", write_neighbors=False)
    torch.cuda.synchronize()
    L.sp_internal_dbg_read(buf, 0)
    h = list(buf)
    print(f"iterations 0..{iters - 1}: slowest single search {h[0] * 0.16:.1f} us; searches over 5 us: {h[1]}, over 10 us: {h[2]}, over 20 us: {h[3]}; lanes with one over 5 us: {h[4]}; workgroup-launches with one over 20 us: {h[5]}", flush=True)
    print(f"    slowest 4x4x4 {h[6] / 100:.1f} us, ball {h[7] / 100:.1f} us, ring walk {h[8] / 100:.1f} us; lanes entering them {h[9]} / {h[10]} / {h[11]}; ball scans over 10 / 20 / 30 us: {h[12]} / {h[13]} / {h[14]}", flush=True)
