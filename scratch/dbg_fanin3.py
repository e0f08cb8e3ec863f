import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, torch, torch.distributed as dist
os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]="29613"; os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY","0")
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
import sycl_points_amd.api as sp
from sycl_points_amd import _lib
from sycl_points_amd.synthetic import gicp_pair
L=_lib.lib()
comm = sp.Communicator.from_process_group(dist.group.WORLD)
n=60000; iters=12
src,tgt,T=gicp_pair(n,10.0*(n/1e6)**(1/3))
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
Tg=sp.PointCloudShared(dev(tgt)); Tg.covs=sp.GridKNN.build(Tg.points,points_per_cell=6.0).self_knn(20,want_knn=False,want_covs=True)[1]
S=sp.PointCloudShared(dev(src)); S.covs=sp.GridKNN.build(S.points,points_per_cell=6.0).self_knn(20,want_knn=False,want_covs=True)[1]
S=S.reordered(sp.GridKNN.build(S.points,points_per_cell=1.0).order())
prep=sp.PreparedTarget(sp.GridKNN.build(Tg.points,points_per_cell=0.5),Tg.covs)
p=sp.RegistrationParams(criteria_translation=0.0,criteria_rotation=0.0,max_iterations=iters)
ident=torch.eye(4,device="cuda").reshape(-1).contiguous()
for variant in ("base", "sync", "mode1", "noallreduce", "mode0"):
  single_call=False
  for sortmode in ("presorted",):
    reg=sp.Registration(p); T_dev=ident.clone(); delta=torch.zeros(8,device="cuda")
    reg.align_fused_loop(S,prep,iterations=iters,T_dev=T_dev,delta_dev=delta,sort_by_cell=sortmode); torch.cuda.synchronize(); ref=T_dev.cpu().numpy().copy()
    ws,lin=reg._buffers(S.points.device); fp=reg._factor_params(10.0); gn=_lib.GnParams(1.0,0.0,0.0)
    itd=torch.zeros(1,dtype=torch.int32,device="cuda")
    def enqueue():
        st=sp._stream()
        if single_call:
            _lib.check(L.sp_gicp_align_sharded(prep._h,reg._psrc._h,sp._ptr(T_dev),C.byref(fp),C.byref(gn),iters,comm._h,None,None,sp._ptr(lin),sp._ptr(delta),sp._ptr(itd),sp._ptr(ws),ws.numel(),st)); return
        mode = 1 if variant=="mode1" else (0 if variant=="mode0" else 2)
        for k in range(iters):
            _lib.check(L.sp_gicp_align_step(prep._h,reg._psrc._h,sp._ptr(T_dev),C.byref(fp),C.byref(gn),k,mode,None,None,sp._ptr(lin),sp._ptr(ws),ws.numel(),st))
            if variant in ("base","sync"): _lib.check(L.sp_allreduce_rows(comm._h,sp._ptr(ws),k,st))
        _lib.check(L.sp_gicp_align_finish(reg._psrc._h,sp._ptr(T_dev),C.byref(gn),iters-1,mode,sp._ptr(lin),sp._ptr(delta),sp._ptr(itd),sp._ptr(ws),ws.numel(),st))
    out=[]
    T_dev.copy_(ident); reg._psrc.prepare(prep,S,T_dev,sortmode); enqueue(); torch.cuda.synchronize(); out.append(float(np.abs(T_dev.cpu().numpy()-ref).max()))
    T_dev.copy_(ident); reg._psrc.prepare(prep,S,T_dev,sortmode); torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): enqueue()
    for r in range(4):
        T_dev.copy_(ident); reg._psrc.prepare(prep,S,T_dev,sortmode)
        if variant=="sync": torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize(); out.append(float(np.abs(T_dev.cpu().numpy()-ref).max()))
    nlog=C.c_size_t(0); lp=L.sp_internal_align_searched_log(sp._ptr(ws),C.byref(nlog)); off=lp-ws.data_ptr()
    print(variant,"sort",sortmode,out,"iters",int(itd[0]),"searched",ws[off:off+4*iters].view(torch.int32).cpu().tolist(),flush=True)
