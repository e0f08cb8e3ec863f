import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
P = torch.from_numpy(Mt19937Cloud(1234).uniform_points(1000000, 10.0)).cuda()
for ppc in (2.0, 3.0, 4.0, 5.0, 6.0, 8.0, 12.0):
    g = sp.GridKNN.build(P, points_per_cell=ppc)
    print("ppc %.1f  k=20 knn %.3f ms  knn+cov %.3f ms  k=10 knn %.3f ms" % (ppc, timed(lambda: g.self_knn(20, True, False, False)), timed(lambda: g.self_knn(20, False, True, False)), timed(lambda: g.self_knn(10, True, False, False))), flush=True)
