import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sycl_points_amd.api as sp
from oracle.pyoracle import Oracle
orc = Oracle()
g = orc.rng(31)
for n, r in ((20000, 5.0), (3000, 40.0)):   # dense-ish and very sparse (ring expansion, to-do)
    pts = g.uniform_points(n, r)
    dup = np.concatenate([pts, pts[:500]])  # duplicates: exact ties
    for cloud in (pts, dup):
        grid = sp.GridKNN.build(torch.from_numpy(cloud).cuda(), points_per_cell=6.0)
        for k in (11, 13, 16, 19, 20):
            res, covs, _ = grid.self_knn(k, want_knn=True, want_covs=True)
            oi, od = orc.knn_bruteforce(cloud, cloud, k)
            ok = np.array_equal(res.indices.cpu().numpy(), oi) and np.array_equal(res.distances.cpu().numpy(), od)
            print(len(cloud), r, k, "OK" if ok else "MISMATCH")
            assert ok
print("all good")
