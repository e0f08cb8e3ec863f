import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud
P = torch.from_numpy(Mt19937Cloud(1234).uniform_points(1000000, 10.0)).cuda()
for _ in range(5): g = sp.GridKNN.build(P, points_per_cell=0.5)
torch.cuda.synchronize(); print(len(P))
