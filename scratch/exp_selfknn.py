import sys; sys.path.insert(0,'.')
import numpy as np, torch
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud
def timed(fn,reps=5):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps*1e3
pts=Mt19937Cloud(1234).uniform_points(1000000,10.0); P=torch.from_numpy(pts).cuda()
tree=sp.KDTree.build(pts); r=sp.KNNResult()
print("kdtree k=20 self: %.0f us ; K5 cov: %.0f us"%(timed(lambda: tree.knn_search_async(P,20,r)), timed(lambda: sp.covariance.estimate(r,P))))
for ppc in (3.0,4.0,5.0,6.0,8.0,12.0):
    g=sp.GridKNN.build(P,points_per_cell=ppc)
    a=timed(lambda: g.self_knn(20,True,False,False)); b=timed(lambda: g.self_knn(20,False,True,False)); c=timed(lambda: g.self_knn(20,True,True,True))
    print("grid ppc %.0f h=%.3f: self kNN20 %.0f us | fused cov only %.0f us | knn+cov+normals %.0f us"%(ppc,g.cell_size(),a,b,c),flush=True)
g=sp.GridKNN.build(P,points_per_cell=4.0)
print("k=10 ppc4: %.0f us"%timed(lambda: g.self_knn(10,True,False,False)))
