import sys; sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
import sycl_points_amd.api as sp
from sycl_points_amd import _lib
from sycl_points_amd.synthetic import gicp_pair
def timed(fn,reps=30):
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
for ppc in [float(a) for a in sys.argv[1:]] or [0.5]:
    grid=sp.GridKNN.build(Tg.points,points_per_cell=ppc); prep=sp.PreparedTarget(grid,Tg.covs)
    reg=sp.Registration(p)
    Tid=dev(np.eye(4,dtype=np.float32).reshape(-1)); delta=torch.zeros(8,device='cuda')
    poses=[]
    for its in (0,1,2,20):
        Td=Tid.clone()
        if its: reg.align_fused_loop(S,prep,iterations=its,T_dev=Td,delta_dev=delta)
        else: reg.align_fused_loop(S,prep,iterations=1,T_dev=Td.clone(),delta_dev=delta)
        poses.append((its,Td))
    ws,lin=reg._buffers(S.points.device); fp=reg._factor_params(10.0)
    def it(Td): _lib.check(L.sp_gicp_iteration_fused(prep._h,reg._psrc._h,sp._ptr(Td),1,C.byref(fp),None,None,None,sp._ptr(lin),sp._ptr(delta),sp._ptr(ws),ws.numel(),sp._stream()))
    for its,Td in poses:
        out=[]
        for mask,name in ((1,"full"),(5,"NN staged"),(17,"NN fast+ring"),(9,"math only")):
            reg._set_source_option("stage_mask", mask)
            out.append("%s %.1f"%(name,timed(lambda: it(Td))))
        reg._set_source_option("stage_mask", 1)
        iters_dev=torch.zeros(1,dtype=torch.int32,device='cuda'); gn=_lib.GnParams(1.0,0.0,0.0)
        def al(Td=Td): _lib.check(L.sp_gicp_align_fused(prep._h,reg._psrc._h,sp._ptr(Td),C.byref(fp),C.byref(gn),1,None,None,sp._ptr(lin),sp._ptr(delta),sp._ptr(iters_dev),sp._ptr(ws),ws.numel(),sp._stream()))
        out.append("align-kernel(no prologue) %.1f"%timed(al))
        print("ppc %.2f pose after %2d iterations: "%(ppc,its)+" | ".join(out)+" us",flush=True)
    reg._set_source_option("stage_mask", 3)
