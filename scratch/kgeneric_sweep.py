import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud
pts=torch.from_numpy(Mt19937Cloud(1234).uniform_points(1000000,10.0)).cuda()
def t(fn,reps=3):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
for ppc in (2.0,6.0):
    g=sp.GridKNN.build(pts,points_per_cell=ppc)
    sorted_pts=pts[g.order()].contiguous()
    for k in (3,6,10,20):
        print("ppc %.1f k=%2d generic on cell-ordered queries %.3f ms | self_knn %.3f ms"%(ppc,k,t(lambda: g.knn_search(sorted_pts,k)),t(lambda: g.self_knn(k,want_knn=True))))
