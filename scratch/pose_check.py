"""Final pose / iteration count of the bench's alignment under the library given by SP_AMD_LIB (A/B of two builds)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import gicp_pair
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
src, tgt, T_gt = gicp_pair(n, 10.0 * (n / 1e6) ** (1 / 3))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
Tg = sp.PointCloudShared(dev(tgt))
Tg.covs = sp.GridKNN.build(Tg.points, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
S_all = dev(src)
S_all = S_all[sp.GridKNN.build(S_all, points_per_cell=1.0).order()].contiguous()
covs = sp.GridKNN.build(S_all, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
S = sp.PointCloudShared(S_all, covs=covs)
prep = sp.PreparedTarget(sp.GridKNN.build(Tg.points, points_per_cell=0.5), Tg.covs)
for crit, opts in ((0.0, {}), (1e-3, {}), (1e-3, {"persistent": 0}), (1e-3, {"persistent_from": 0}), (1e-3, {"persistent_from": 1}),
                   (1e-3, {"persistent_from": 2}), (1e-3, {"persistent_from": 3}), (1e-3, {"persistent_from": 7}), (1e-7, {"persistent_from": 2})):
    for wn in (False, True):
        p = sp.RegistrationParams(criteria_translation=crit, criteria_rotation=crit, max_iterations=20)
        reg = sp.Registration(p)
        for k_, v_ in opts.items():
            reg._set_source_option(k_, v_)
        T_dev, lin, delta = reg.align_fused_loop(S, prep, sort_by_cell="presorted", write_neighbors=wn)
        torch.cuda.synchronize()
        T = T_dev.cpu().numpy().reshape(4, 4).T
        print(os.environ.get("SP_AMD_LIB", "default")[-20:], "crit", crit, opts, "wn", wn, "iters", int(reg._iters_dev[0]),
              "err vs gt %.3e" % np.abs(T - T_gt).max(), "inliers", reg._read_lin(lin).inlier, "T00 %.9f t0 %.9f" % (T[0, 0], T[0, 3]), "delta6", float(delta[6]), "lin.err %.6e" % reg._read_lin(lin).error)
