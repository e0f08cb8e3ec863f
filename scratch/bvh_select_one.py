import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sycl_points_amd.api as sp
from test_gpu_bvh import nonuniform_cloud
P = torch.from_numpy(nonuniform_cloud(1_000_000)).cuda()
b = sp.BVH.build(P)
for i in range(3):
    r = b.self_knn(20)
torch.cuda.synchronize()
