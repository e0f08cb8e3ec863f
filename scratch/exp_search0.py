"""Launch-0 workload in isolation: exact NN (k = 1) of every source point at the identity pose, cell-ordered queries."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import gicp_pair
n=1000000
src,tgt,T=gicp_pair(n,10.0)
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
Tg=dev(tgt); S=dev(src)
S=S[sp.GridKNN.build(S,points_per_cell=1.0).order()].contiguous()
def timed(fn,reps=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps*1e3
res=sp.KNNResult()
for ppc in (0.5,1.0,2.0):
    g=sp.GridKNN.build(Tg,points_per_cell=ppc)
    q=sp.PointCloudShared(S)
    print("ppc %.1f  grid_search k=1 (identity pose, cell-ordered queries): %.1f us"%(ppc,timed(lambda: g.knn_search_async(q,1,res))),flush=True)
    qr=sp.PointCloudShared(dev(src))
    print("ppc %.1f  grid_search k=1 (random-order queries): %.1f us"%(ppc,timed(lambda: g.knn_search_async(qr,1,res))),flush=True)
