"""full-resolution alignment of the reference's bundled scans (69 792 vs 69 088 points, no sampling, no
voxel filter) — density varying by orders of magnitude — through the in-loop searches the library has:
  grid (volume rule) | grid (cell size steered by occupancy, sp_grid_create_adaptive) | device-built hierarchy (generic loop:
  sp_bvh_search k = 1 + K11 per iteration). ms per Gauss-Newton iteration and the pose against the oracle's."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sycl_points_amd.api as sp
from oracle.pyoracle import Oracle, RegParams
from test_gpu_facade import GOLD, read_ply_xyz

orc = Oracle()
src, tgt = read_ply_xyz(os.path.join(GOLD, "source.ply")), read_ply_xyz(os.path.join(GOLD, "target.ply"))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
S, Tg = sp.PointCloudShared(dev(src)), sp.PointCloudShared(dev(tgt))
for c in (S, Tg):
    sp.covariance.estimate(sp.BVH.build(c.points).self_knn(20), c)
scov, tcov = S.covs.cpu().numpy(), Tg.covs.cpu().numpy()
ITERS = 20
p = sp.RegistrationParams(max_iterations=ITERS, criteria_rotation=0.0, criteria_translation=0.0)
ref = orc.registration_align(RegParams.defaults(max_iterations=ITERS, crit_rotation=0.0, crit_translation=0.0), src, scov, tgt, tcov)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms)), out

for name, grid in (("grid, volume rule", sp.GridKNN.build(Tg.points, points_per_cell=0.5)),
                   ("grid, occupancy-steered", sp.GridKNN.build(Tg.points, points_per_cell=0.5, adaptive=True))):
    prep = sp.PreparedTarget(grid, Tg.covs)
    reg = sp.Registration(p)
    ms, (T_dev, lin, delta) = timed(lambda: reg.align_fused_loop(S, prep, sort_by_cell=True))
    T = reg.T_from_device(T_dev)
    print(f"{name:28s} cell {grid.cell_size():.3f} m, fullest cell {grid.max_cell_points():5d}: {ms / ITERS * 1e3:8.1f} us per iteration "
          f"({ms:.2f} ms per alignment); |T - T_oracle| = {np.abs(T - ref['T']).max():.2e}, inliers {reg._read_lin(lin).inlier} / {ref['inlier']}")
bvh = sp.BVH.build(Tg.points)
reg = sp.Registration(p)
ms, (T_dev, lin, delta) = timed(lambda: reg.align_device_loop(S, Tg, bvh, iterations=ITERS), 3)
T = reg.T_from_device(T_dev)
print(f"{'hierarchy (generic loop)':28s} {'':41s}{ms / ITERS * 1e3:8.1f} us per iteration ({ms:.2f} ms per alignment); "
      f"|T - T_oracle| = {np.abs(T - ref['T']).max():.2e}")

# the device-resident optimiser loop (Gauss-Newton), one wave per source point when the library
# is told to (sp_gicp_source_set_wave_per_point = 2: what the facade does for a target with crowded cells), a lane per point otherwise
for name, grid in (("grid, volume rule", sp.GridKNN.build(Tg.points, points_per_cell=0.5)),
                   ("grid, occupancy-steered", sp.GridKNN.build(Tg.points, points_per_cell=0.5, adaptive=True)),
                   ("grid, 0.3 m cells", sp.GridKNN.build(Tg.points, cell_size=0.3)),
                   ("grid, 0.15 m cells", sp.GridKNN.build(Tg.points, cell_size=0.15))):
    prep = sp.PreparedTarget(grid, Tg.covs)
    print(name, "cell", round(grid.cell_size(), 3), "fullest", grid.max_cell_points())
    for wq in (2, 0):
        reg = sp.Registration(sp.RegistrationParams(max_iterations=ITERS, criteria_rotation=0.0, criteria_translation=0.0,
                                                    optimization_method="GN"))
        reg._set_source_option("opt_wave_query", wq)
        ms, res = timed(lambda: reg.align_optimize(S, prep, None, None, True))
        print(f"optimiser launch, GN, opt_wave_query={wq}: {ms / ITERS * 1e3:8.1f} us per iteration ({ms:.2f} ms per alignment); "
              f"|T - T_oracle| = {np.abs(res.T - ref['T']).max():.2e}, inliers {res.inlier} / {ref['inlier']}, searched {res.searched}")
