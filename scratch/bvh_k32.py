import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sycl_points_amd.api as sp
from test_gpu_bvh import nonuniform_cloud
from sycl_points_amd.synthetic import Mt19937Cloud
def med(fn, runs=5):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(runs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1))
    return float(np.median(t))
for name, pts in (("non-uniform", nonuniform_cloud(1_000_000)), ("uniform", Mt19937Cloud(1234).uniform_points(1_000_000, 10.0))):
    P = torch.from_numpy(pts).cuda()
    b = sp.BVH.build(P)
    r = sp.KNNResult()
    for k in (21, 24, 32):
        print(name, k, "self", round(med(lambda: b.self_knn(k)), 3), "external", round(med(lambda: b.knn_search_async(P, k, r)), 3), flush=True)
