import sys; sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
import sycl_points_amd.api as sp
from sycl_points_amd import _lib
from sycl_points_amd.synthetic import gicp_pair
def timed(fn,reps=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps*1e3
n=1000000
src,tgt,T=gicp_pair(n,10.0)
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
Tg=sp.PointCloudShared(dev(tgt)); S=sp.PointCloudShared(dev(src))
gk=sp.GridKNN.build(Tg.points,points_per_cell=8.0); gs=sp.GridKNN.build(S.points,points_per_cell=8.0)
sp.covariance.estimate(gk.knn_search(Tg,20),Tg); sp.covariance.estimate(gs.knn_search(S,20),S)
p=sp.RegistrationParams(criteria_translation=0.0,criteria_rotation=0.0,max_iterations=20)
L=_lib.lib()
grid=sp.GridKNN.build(Tg.points,points_per_cell=0.5); prep=sp.PreparedTarget(grid,Tg.covs)
Tid=dev(np.eye(4,dtype=np.float32).reshape(-1)); delta=torch.zeros(8,device='cuda')
def cell_order(P, ppc):
    x=P[:,:3]; lo=x.min(0).values; hi=x.max(0).values
    h=float(((hi-lo).prod()/ (P.shape[0]/ppc))**(1/3))
    c=((x-lo)/h).floor().long(); d=((hi-lo)/h).floor().long()+1
    key=(c[:,2]*d[1]+c[:,1])*d[0]+c[:,0]
    return torch.argsort(key,stable=True)
def run(name,Sx,sort,fast):
    reg=sp.Registration(p)
    reg._set_source_option("fast_nn", fast)
    def one():
        Td=Tid.clone(); reg.align_fused_loop(Sx,prep,iterations=20,T_dev=Td,delta_dev=delta,sort_by_cell=sort); return Td
    Td=one(); err=np.abs(reg.T_from_device(Td)-T).max()
    t=timed(one)
    tp=timed(lambda: reg._psrc.prepare(prep,Sx,Tid,sort))
    print("%-40s alignment %.0f us (%.1f/iter) prepare %.0f us err %.1e"%(name,t,t/20,tp,err),flush=True)
run("cell-sorted (baseline)",S,True,-1)
run("random order, fast nn",S,False,1)
run("random order, ring nn",S,False,0)
for ppc in (8.0,2.0,0.5):
    perm=cell_order(S.points,ppc)
    S2=sp.PointCloudShared(S.points[perm].contiguous(),covs=S.covs[perm].contiguous())
    run("source-grid order ppc %.1f, fast nn"%ppc,S2,False,1)
reg._set_source_option("fast_nn", -1)
