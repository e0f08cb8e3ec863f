import os, sys; sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
import sycl_points_amd.api as sp
from sycl_points_amd import _lib
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
Tg=sp.PointCloudShared(dev(tgt)); S0=sp.PointCloudShared(dev(src))
S=S0.reordered(sp.GridKNN.build(S0.points,points_per_cell=1.0).order())
gk=sp.GridKNN.build(Tg.points,points_per_cell=8.0); gs=sp.GridKNN.build(S.points,points_per_cell=8.0)
sp.covariance.estimate(gk.knn_search(Tg,20),Tg); sp.covariance.estimate(gs.knn_search(S,20),S)
p=sp.RegistrationParams(criteria_translation=0.0,criteria_rotation=0.0,max_iterations=20)
L=_lib.lib()
grid=sp.GridKNN.build(Tg.points,points_per_cell=0.5); prep=sp.PreparedTarget(grid,Tg.covs)
reg=sp.Registration(p)
Tid=dev(np.eye(4,dtype=np.float32).reshape(-1)); delta=torch.zeros(8,device='cuda')
Td=Tid.clone(); reg.align_fused_loop(S,prep,iterations=20,T_dev=Td,delta_dev=delta,sort_by_cell="presorted")
ws,lin=reg._buffers(S.points.device); fp=reg._factor_params(10.0); gn=_lib.GnParams(1.0,0.0,0.0)
it=torch.zeros(1,dtype=torch.int32,device='cuda')
reg._set_source_option("stage_mask", 1)
def run(k):
    def f():
        Tc=Td.clone()
        _lib.check(L.sp_gicp_align_fused(prep._h,reg._psrc._h,sp._ptr(Tc),C.byref(fp),C.byref(gn),k,None,None,sp._ptr(lin),sp._ptr(delta),sp._ptr(it),sp._ptr(ws),ws.numel(),sp._stream()))
    return f
t=[timed(run(k)) for k in (1,2,3,5,9)]
print("launch counts 1,2,3,5,9 at the converged pose: %s us"%[round(x,1) for x in t])
print("per extra launch (with prologue): %.1f us ; first launch (no prologue): %.1f us"%((t[4]-t[0])/8, t[0]))
reg._set_source_option("stage_mask", 3)
if os.environ.get("SP_TM"):
    import struct
    reg._set_source_option("stage_mask", 1)
    run(4)(); torch.cuda.synchronize()
    raw = ws.cpu().numpy().tobytes()
    off = 2 * 256 * 32 * 4 + 2 * 112
    t = struct.unpack("34Q", raw[off:off + 34 * 8])
    print("block 0, ticks from kernel-side start: prologue done %d ; per-wave main-loop end %s ; block reduce done %d" % (t[1] - t[0], [t[17 + w] - t[0] for w in range(16)], t[33] - t[0]))

if os.environ.get("SP_TM0"):
    import struct
    reg._set_source_option("stage_mask", 1)
    reg._psrc.prepare(prep, S, Tid, "presorted")
    Tc = Tid.clone()
    _lib.check(L.sp_gicp_align_fused(prep._h,reg._psrc._h,sp._ptr(Tc),C.byref(fp),C.byref(gn),1,None,None,sp._ptr(lin),sp._ptr(delta),sp._ptr(it),sp._ptr(ws),ws.numel(),sp._stream()))
    torch.cuda.synchronize()
    raw = ws.cpu().numpy().tobytes()
    off = 2 * 256 * 32 * 4 + 2 * 112
    t = struct.unpack("34Q", raw[off:off + 34 * 8])
    print("first launch (identity pose), block 0 ticks: loop start %s ; per-wave loop end %s ; reduce done %d" % ([t[1 + w] - t[0] for w in (0, 15)], [t[17 + w] - t[0] for w in range(16)], t[33] - t[0]))
