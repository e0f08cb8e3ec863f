import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud
k = int(sys.argv[1]); mode = int(sys.argv[2])
P = torch.from_numpy(Mt19937Cloud(1234).uniform_points(1_000_000, 10.0)).cuda()
grid = sp.GridKNN.build(P, points_per_cell=6.0)
grid._set_option("self_knn_mode", mode)
for _ in range(5):
    grid.self_knn(k, False, True, False)
torch.cuda.synchronize()
