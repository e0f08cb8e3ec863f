import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud
P = torch.from_numpy(Mt19937Cloud(1234).uniform_points(1000000, 10.0)).cuda()
vg = sp.VoxelGrid(0.1)
for _ in range(5): out = vg.downsampling(P)
torch.cuda.synchronize(); print(out.size())
