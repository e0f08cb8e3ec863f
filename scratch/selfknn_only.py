import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud
g=Mt19937Cloud(1234)
pts=torch.from_numpy(g.uniform_points(1000000,10.0)).cuda()
grid=sp.GridKNN.build(pts,points_per_cell=float(sys.argv[1]) if len(sys.argv)>1 else 6.0)
for _ in range(3):
    grid.self_knn(20,want_knn=False,want_covs=True)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): grid.self_knn(20,want_knn=False,want_covs=True)
e1.record(); torch.cuda.synchronize(); print("self_knn+cov ms",e0.elapsed_time(e1)/5)
