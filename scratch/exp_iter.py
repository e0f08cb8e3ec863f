import sys; sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
import sycl_points_amd.api as sp
from sycl_points_amd import _lib
from sycl_points_amd.synthetic import gicp_pair
def timed(fn,reps=50):
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
ppc=float(sys.argv[1]) if len(sys.argv)>1 else 0.5
grid=sp.GridKNN.build(Tg.points,points_per_cell=ppc); prep=sp.PreparedTarget(grid,Tg.covs)
reg=sp.Registration(p)
Tid=dev(np.eye(4,dtype=np.float32).reshape(-1)); Td=Tid.clone(); delta=torch.zeros(8,device='cuda')
reg.align_fused_loop(S,prep,iterations=20,T_dev=Td,delta_dev=delta)
ws,lin=reg._buffers(S.points.device); fp=reg._factor_params(10.0); gn=_lib.GnParams(1.0,0.0,0.0)
def it(gnp): _lib.check(L.sp_gicp_iteration_fused(prep._h,reg._psrc._h,sp._ptr(Td),1,C.byref(fp),gnp,None,None,sp._ptr(lin),sp._ptr(delta),sp._ptr(ws),ws.numel(),sp._stream()))
print("converged: iteration w/o solve %.1f us ; with fused solve %.1f us"%(timed(lambda: it(None)), timed(lambda: it(C.byref(gn)))))
print("gn_update alone %.1f us"%timed(lambda: _lib.check(L.sp_gn_update(sp._ptr(lin),sp._ptr(Td),1.0,0.0,0.0,sp._ptr(delta),sp._stream()))))
print("prepare: sorted %.1f us unsorted %.1f us ; target update %.1f us"%(timed(lambda: reg._psrc.prepare(prep,S,Tid,True)),timed(lambda: reg._psrc.prepare(prep,S,Tid,False)),timed(lambda: prep.update())))
reg._psrc.prepare(prep,S,Tid,True)
def one_align():
    Td.copy_(Tid); reg.align_fused_loop(S,prep,iterations=20,T_dev=Td,delta_dev=delta)
t=timed(one_align,reps=10); print("alignment %.0f us = %.1f us/iter, err %.1e"%(t,t/20,np.abs(reg.T_from_device(Td)-T).max()))
for mask,name in ((1,"full kernel"),(5,"NN only"),(9,"math+stream only (no search)")):
    reg._set_source_option("stage_mask", mask)
    print("%-32s %.1f us"%(name, timed(lambda: it(None))))
reg._set_source_option("stage_mask", 3)
