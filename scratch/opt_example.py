"""The registration stage of the reference's example (LM + Geman-McClure + 3 annealing levels on a 1000-point sample of the
bundled scans, downsampled at 0.25 m) through sp_gicp_align_optimize: time of the launch and the steps it ran, under the
switches that decide where its time goes. GPU box: python scratch/opt_example.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sycl_points_amd.api as sp  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402  (clouds only: the CPU restatement of the example's preprocessing)
from test_gpu_facade import GOLD, read_ply_xyz  # noqa: E402

orc = Oracle()


def prep(p):
    p = p[orc.box_filter(p, 0.5, 50.0) == 1]
    p = orc.voxel_downsample(p, 0.25, 1, stable=True)["points"]
    idx, _ = orc.kdtree_knn(orc.kdtree_build(p), p, 10)
    return p, orc.cov_estimate(p, idx)


s, sc = prep(read_ply_xyz(os.path.join(GOLD, "source.ply")))
t, tc = prep(read_ply_xyz(os.path.join(GOLD, "target.ply")))
keep = orc.random_sampling_flags(1234, len(s), 1000) == 1
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
S = sp.PointCloudShared(dev(s[keep]), covs=dev(sc[keep]))
Sall = sp.PointCloudShared(dev(s), covs=dev(sc))
Tg = sp.PointCloudShared(dev(t), covs=dev(tc))
scales = [10.0, 5.0, 2.5]


def run(label, grid, src, sort, opt="LM", max_it=10, levels=scales, reuse=None, reps=20):
    prep_t = sp.PreparedTarget(grid, Tg.covs)
    p = sp.RegistrationParams(robust_type="GEMAN_MCCLURE", optimization_method=opt, max_iterations=max_it)
    reg = sp.Registration(p)
    reg._prepared_source(src.size())
    if reuse is not None:
        reg._set_source_option("reuse", reuse)
    ms = []
    for _ in range(reps):
        reg._psrc.prepare(prep_t, src, None, sort)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        res_dev = reg.align_optimize(src, prep_t, None, levels, sort, prepare=False, enqueue_only=True)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    r = sp._lib.AlignResult.from_buffer_copy(res_dev.cpu().numpy().tobytes())
    print(f"{label:58s} {1e3 * np.median(ms[3:]):8.1f} us  lin {r.linearizations:3d} trials {r.trials:3d} searched {r.searched:7d} "
          f"cell {grid.cell_size():.3f} maxcell {grid.max_cell_points()}")


g_plain = sp.GridKNN.build(Tg.points, points_per_cell=0.5)
g_adapt = sp.GridKNN.build(Tg.points, points_per_cell=0.5, adaptive=True)
for name, g in (("plain grid", g_plain), ("adaptive grid", g_adapt)):
    run(f"{name}, 1000 pts, sorted by cell", g, S, True)
    run(f"{name}, 1000 pts, as sampled (ring walk)", g, S, False)
    run(f"{name}, 1000 pts, reuse off", g, S, True, reuse=0)
    run(f"{name}, 1000 pts, 1 level 1 iteration", g, S, True, max_it=1, levels=[10.0])
    run(f"{name}, 1000 pts, 1 level 2 iterations", g, S, True, max_it=2, levels=[10.0])
    run(f"{name}, 1000 pts, GN 3 levels", g, S, True, opt="GN")
    run(f"{name}, all {Sall.size()} pts, sorted", g, Sall, True)
for h in (0.3, 0.5, 0.75, 1.0, 1.5):
    g = sp.GridKNN.build(Tg.points, cell_size=h)
    run(f"cell {h}, 1000 pts, sorted by cell", g, S, True)

if os.environ.get("SP_OPT_TIMING"):  # a library built with -DSP_OPT_TIMING: stamps of the first steps (100 MHz ticks)
    import ctypes as C
    which = os.environ["SP_OPT_TIMING"]
    g = {"plain": g_plain, "adaptive": g_adapt}.get(which) or sp.GridKNN.build(Tg.points, cell_size=float(which))
    prep_t = sp.PreparedTarget(g, Tg.covs)
    reg = sp.Registration(sp.RegistrationParams(robust_type="GEMAN_MCCLURE", optimization_method="LM", max_iterations=10))
    for _ in range(3):
        res_dev = reg.align_optimize(S, prep_t, None, scales, True, enqueue_only=True)
        torch.cuda.synchronize()
    raw = res_dev.cpu().numpy().tobytes()
    off = sp._lib.AlignResult.log.offset + 8 * 16
    st = np.frombuffer(raw[off:off + 800], dtype=np.uint64).reshape(20, 5).astype(np.int64)
    dbg = np.zeros(24 * 16, np.uint64)
    L = C.CDLL(sp._lib.LIB_PATH)
    L.sp_internal_opt_debug(dbg.ctypes.data_as(C.c_void_p))
    dbg = dbg.astype(np.int64).reshape(24, 16)
    wg = np.zeros(24 * 8, np.uint64)
    L.sp_internal_opt_debug_wg(wg.ctypes.data_as(C.c_void_p))
    wg = wg.astype(np.int64).reshape(24, 8) / 100.0
    print(f"grid {which}: cell {g.cell_size():.3f}")
    print("points loop + workgroup reduction per workgroup (us), steps 0..18:")
    for k in range(19):
        print(f"  {k:2d}: " + " ".join(f"{v:6.1f}" for v in wg[k][:4]))
    print("step: points | block reduce | rows/barrier | state machine  ||  wave 0: loads+cert | fast/seeded | 4x4x4 | ball(wave) | store | math || lanes: searching seeded past-fast open")
    for k in range(20):
        a = st[k]
        if a[1] == 0:
            break
        pts_end = a[4] if a[4] else a[1]
        d = dbg[k]
        extra = ""
        if a[4]:
            extra = " || %5.2f | %5.2f | %5.2f | %5.2f | %5.2f | %5.2f || %2d %2d %2d %2d" % (
                (d[1] - d[0]) / 100, (d[2] - d[1]) / 100, (d[3] - d[2]) / 100, (d[4] - d[3]) / 100, (d[5] - d[4]) / 100,
                (d[6] - d[5]) / 100, d[8], d[11], d[9], d[10])
        print(f"{k:2d}: {(pts_end-a[0])/100:7.2f} | {(a[1]-pts_end)/100:7.2f} | {(a[2]-a[1])/100:7.2f} | {(a[3]-a[2])/100:7.2f}{extra}")
