"""where a searching launch spends its time: later search stages inside gicp_align_kernel, launch by launch
(git apply scratch/experiments/r04_search_stage_stamps.patch; bash scratch/devbuild.sh -DSP_SEARCH_DBG). usage: search_dbg.py [world]  (emulated rank 0 of `world` ranks)"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sycl_points_amd.api as sp
from sycl_points_amd.sharding import shard_indices
from sycl_points_amd.synthetic import gicp_pair
import test_gpu_persistent_tail as tp
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = 1_000_000 * world
src, tgt, T_gt = gicp_pair(n, 10.0 * world ** (1.0 / 3.0))
dev = tp.dev
Tg = sp.PointCloudShared(dev(tgt))
Tg.covs = sp.GridKNN.build(Tg.points, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
S_all = dev(src)
S_all = S_all[sp.GridKNN.build(S_all, points_per_cell=1.0).order()].contiguous()
covs = sp.GridKNN.build(S_all, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
tile = torch.from_numpy(shard_indices(n, 0, world, 1024)).cuda()
S = sp.PointCloudShared(S_all[tile].contiguous(), covs=covs[tile].contiguous())
prep = sp.PreparedTarget(sp.GridKNN.build(Tg.points, points_per_cell=0.5), Tg.covs)
L = sp._lib.lib()
L.sp_internal_dbg_read.restype = C.c_int
buf = (C.c_uint * 16)()
prev = [0] * 16
for iters in range(1, 7):
    L.sp_internal_dbg_read(None, 1)
    p = sp.RegistrationParams(criteria_translation=0.0, criteria_rotation=0.0, max_iterations=iters)
    reg = sp.Registration(p)
    reg._set_source_option("persistent", 0)
    reg.align_fused_loop(S, prep, sort_by_cell="presorted", write_neighbors=False)
    torch.cuda.synchronize()
    L.sp_internal_dbg_read(buf, 0)
    h = list(buf)
    d = [h[i] - prev[i] for i in range(16)]
    print(f"launch {iters - 1}: lanes entering 4x4x4 / ball / ring walk {d[9]} / {d[10]} / {d[11]}; slowest so far (us) 4x4x4 {h[6] / 100:.1f}, ball {h[7] / 100:.1f}, "
          f"ring walk {h[8] / 100:.1f}; ball scans over 10 / 20 / 30 us: {d[12]} / {d[13]} / {d[14]}; workgroups with a ball scan over 20 us {d[5]}, with a ring walk {d[15]}", flush=True)
    prev = h
