import sys; sys.path.insert(0,'.')
import numpy as np, torch, time
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import gicp_pair
def timed(fn,reps=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps*1e3
n=1000000
src,tgt,T=gicp_pair(n,10.0)
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
Tg=sp.PointCloudShared(dev(tgt)); S=sp.PointCloudShared(dev(src))
tree=sp.KDTree.build(tgt)
Tgt=dev(np.ascontiguousarray(T.T).reshape(-1)); Tid=dev(np.eye(4,dtype=np.float32).reshape(-1))
res=sp.KNNResult()
print("kdtree k=1 identity %.1f us  converged %.1f us"%(timed(lambda: tree.knn_search_async(S,1,res,Tid)), timed(lambda: tree.knn_search_async(S,1,res,Tgt))))
for ppc in (0.25,0.5,1.0,2.0,4.0,8.0):
    torch.cuda.synchronize(); t0=time.time(); grid=sp.GridKNN.build(Tg.points,points_per_cell=ppc); torch.cuda.synchronize(); tb=time.time()-t0
    a=timed(lambda: grid.knn_search_async(S,1,res,Tid)); b=timed(lambda: grid.knn_search_async(S,1,res,Tgt))
    c=timed(lambda: grid.knn_search_async(Tg,20,res),reps=3) if ppc>=2 else float('nan')
    print("grid ppc=%.2f h=%.3f build %.1f ms | k=1 identity %.1f us converged %.1f us | k=20 self %.1f us"%(ppc,grid.cell_size(),tb*1e3,a,b,c),flush=True)
print("kdtree k=20 self %.1f us"%timed(lambda: tree.knn_search_async(Tg,20,res),reps=3))
