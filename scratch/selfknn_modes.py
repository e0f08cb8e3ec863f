import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sycl_points_amd.api as sp
from sycl_points_amd import _lib
from sycl_points_amd.synthetic import Mt19937Cloud
pts=torch.from_numpy(Mt19937Cloud(1234).uniform_points(1000000,10.0)).cuda()
L=_lib.lib()
def t(fn,reps=5):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
for ppc in (2.0,6.0):
    g=sp.GridKNN.build(pts,points_per_cell=ppc)
    for k in (3,6,10):
        out=[]; res=[]
        for mode in (0,1,2):
            g._set_option('self_knn_mode',mode)
            out.append(t(lambda: g.self_knn(k,want_knn=True,want_covs=True)))
            r=g.self_knn(k,want_knn=True,want_covs=True); res.append((r[0].indices.clone(),r[0].distances.clone(),r[1].clone()))
        g._set_option('self_knn_mode',0)
        same=all(torch.equal(res[0][j],res[m][j]) for m in (1,2) for j in range(3))
        print("ppc %.1f k=%2d  kNN+cov: lane %.3f ms  tile %.3f ms  wave %.3f ms   identical: %s"%(ppc,k,out[0],out[1],out[2],same))
