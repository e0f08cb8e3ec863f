import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud
P = torch.from_numpy(Mt19937Cloud(1234).uniform_points(1000000, 10.0)).cuda()
b = sp.BVH.build(P)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if len(sys.argv) > 2 and sys.argv[2] == 'old':
    b._set_option('bvh_self_heap', 0)
for _ in range(3):
    r = b.self_knn(k)
torch.cuda.synchronize()
print(int(r.indices[0, 0]))
