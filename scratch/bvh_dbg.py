import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sycl_points_amd.api as sp
from test_gpu_bvh import nonuniform_cloud
from sycl_points_amd.synthetic import Mt19937Cloud
for name, pts in (("non-uniform", nonuniform_cloud(1_000_000)), ("uniform", Mt19937Cloud(1234).uniform_points(1_000_000, 10.0))):
    P = torch.from_numpy(pts).cuda()
    b = sp.BVH.build(P)
    print(name, flush=True)
    b.self_knn(20); torch.cuda.synchronize()
