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
tree=sp.KDTree.build(tgt); stree=sp.KDTree.build(src)
sp.covariance.estimate(tree.knn_search(Tg,20),Tg); sp.covariance.estimate(stree.knn_search(S,20),S)
p=sp.RegistrationParams(criteria_translation=0.0,criteria_rotation=0.0,max_iterations=20)
L=sp._lib.lib()
for ppc in (0.5,1.0,2.0,4.0):
  grid=sp.GridKNN.build(Tg.points,points_per_cell=ppc)
  prep=sp.PreparedTarget(grid,Tg.covs)
  for sort in (False,True):
    for fast in (0,1):
        reg._set_source_option("fast_nn", fast)
        reg=sp.Registration(p)
        Tid=dev(np.eye(4,dtype=np.float32).reshape(-1)); Td=Tid.clone(); delta=torch.zeros(8,device='cuda')
        def one_align():
            Td.copy_(Tid); reg.align_fused_loop(S,prep,iterations=20,T_dev=Td,delta_dev=delta,sort_by_cell=sort)
        t=timed(one_align,reps=5)
        tp=timed(lambda: reg._psrc.prepare(prep,S,Tid,sort))
        t1=timed(lambda: reg.align_fused_loop(S,prep,iterations=1,T_dev=Td,delta_dev=delta,prepare=False))
        print("ppc %.1f sort %d fast %d: alignment %.0f us = %.1f us/iter | src prepare %.1f us | converged iteration %.1f us | err %.1e"%(ppc,sort,fast,t,t/20,tp,t1,np.abs(reg.T_from_device(Td)-T).max()),flush=True)
reg._set_source_option("fast_nn", -1)
