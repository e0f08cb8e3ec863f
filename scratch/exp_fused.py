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
g20=sp.GridKNN.build(Tg.points,points_per_cell=2.0)
tree=sp.KDTree.build(tgt); stree=sp.KDTree.build(src)
sp.covariance.estimate(tree.knn_search(Tg,20),Tg); sp.covariance.estimate(stree.knn_search(S,20),S)
p=sp.RegistrationParams(criteria_translation=0.0,criteria_rotation=0.0,max_iterations=20)
for ppc in (0.5,1.0,2.0):
    grid=sp.GridKNN.build(Tg.points,points_per_cell=ppc)
    prep=sp.PreparedTarget(grid,Tg.covs)
    reg=sp.Registration(p)
    Tid=dev(np.eye(4,dtype=np.float32).reshape(-1)); Td=Tid.clone(); delta=torch.zeros(8,device='cuda')
    def one_align():
        Td.copy_(Tid); reg.align_fused_loop(S,prep,iterations=20,T_dev=Td,delta_dev=delta)
    t=timed(one_align,reps=5)
    print("ppc %.1f: fused alignment (prep + 20 iterations) %.1f us => %.1f us/iter ; pose err %.2e"%(ppc,t,t/20,np.abs(reg.T_from_device(Td)-T).max()))
    covp=sp.prepare_source_covs(S.covs)
    print("   prep source %.1f us, prep target %.1f us"%(timed(lambda: sp.prepare_source_covs(S.covs,covp)), timed(lambda: prep.update())))
    t1=timed(lambda: reg.align_fused_loop(S,prep,iterations=1,T_dev=Td,delta_dev=delta,src_covp=covp))
    print("   one converged iteration (fused kernel + reduce/solve): %.1f us"%t1)
