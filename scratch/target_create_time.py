import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud
P = torch.from_numpy(Mt19937Cloud(1234).uniform_points(1_000_000, 10.0)).cuda()
covs = sp.GridKNN.build(P, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
grid = sp.GridKNN.build(P, points_per_cell=0.5)
for _ in range(3):
    t = sp.PreparedTarget(grid, covs); del t
torch.cuda.synchronize()
ts = []
for _ in range(9):
    torch.cuda.synchronize(); t0 = time.perf_counter(); t = sp.PreparedTarget(grid, covs); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); del t
print("sp_gicp_target_create, 1 M points (host clock, synchronised): median %.3f ms" % float(np.median(ts)))
