import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
Tg=sp.PointCloudShared(dev(tgt))
gk=sp.GridKNN.build(Tg.points,points_per_cell=6.0)
Tg.covs=gk.self_knn(20,want_knn=False,want_covs=True)[1]
grid=sp.GridKNN.build(Tg.points,points_per_cell=0.5); prep=sp.PreparedTarget(grid,Tg.covs)
L=_lib.lib()
for margin in (0.0, 0.5, 1.0):
    keep=np.abs(src[:,:3]).max(1) < 10.0-margin
    S0=sp.PointCloudShared(dev(src[keep]))
    S=S0.reordered(sp.GridKNN.build(S0.points,points_per_cell=1.0).order())
    S.covs=sp.GridKNN.build(S.points,points_per_cell=6.0).self_knn(20,want_knn=False,want_covs=True)[1]
    p=sp.RegistrationParams(criteria_translation=0.0,criteria_rotation=0.0,max_iterations=20)
    reg=sp.Registration(p)
    Tid=dev(np.eye(4,dtype=np.float32).reshape(-1)); delta=torch.zeros(8,device='cuda')
    reg.align_fused_loop(S,prep,iterations=20,T_dev=Tid.clone(),delta_dev=delta,sort_by_cell="presorted")
    ws,lin=reg._buffers(S.points.device); fp=reg._factor_params(10.0); gn=_lib.GnParams(1.0,0.0,0.0)
    it=torch.zeros(1,dtype=torch.int32,device='cuda')
    reg._set_source_option("stage_mask", 1)
    def run(k):
        def f():
            reg._psrc.prepare(prep,S,Tid,"presorted")
            Tc=Tid.clone()
            _lib.check(L.sp_gicp_align_fused(prep._h,reg._psrc._h,sp._ptr(Tc),C.byref(fp),C.byref(gn),k,None,None,sp._ptr(lin),sp._ptr(delta),sp._ptr(it),sp._ptr(ws),ws.numel(),sp._stream()))
        return f
    t0=timed(run(0) if False else (lambda: (reg._psrc.prepare(prep,S,Tid,"presorted"), Tid.clone())))
    t1=timed(run(1)); t2=timed(run(2))
    reg._set_source_option("stage_mask", 3)
    ns=int(keep.sum())
    print("source margin %.1f m inside the target box: %d points ; launch 0: %.1f us (%.1f us per 1M points) ; launch 1: %.1f us (%.1f per 1M)"%(margin,ns,t1-t0,(t1-t0)*1e6/ns,t2-t1,(t2-t1)*1e6/ns),flush=True)
