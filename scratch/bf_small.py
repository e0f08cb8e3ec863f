"""kNN on small clouds (the reference example's 6 k-point downsampled scans): brute force against hierarchy build + search."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud

def med(fn, n=15):
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return 1e3 * float(np.median(ms))

for n in (1000, 3000, 6000, 12000, 16000, 24000, 32000):
    P = torch.from_numpy(Mt19937Cloud(7).uniform_points(n, 10.0)).cuda()
    for k in (1, 10, 20):
        bf = med(lambda: sp.knn_search_bruteforce(P, P, k))
        bvh_build = med(lambda: sp.BVH.build(P), 7)
        b = sp.BVH.build(P)
        bvh = med(lambda: b.self_knn(k))
        print(f"n {n:6d} k {k:2d}: brute force {bf:8.1f} us | hierarchy build {bvh_build:7.1f} + self-kNN {bvh:7.1f} us")
