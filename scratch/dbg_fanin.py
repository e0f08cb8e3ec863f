import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]="29611"; os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY","0")
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import gicp_pair
comm = sp.Communicator.from_process_group(dist.group.WORLD)
for n in (60000,):
    src,tgt,T=gicp_pair(n,10.0*(n/1e6)**(1/3))
    dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    Tg=sp.PointCloudShared(dev(tgt)); Tg.covs=sp.GridKNN.build(Tg.points,points_per_cell=6.0).self_knn(20,want_knn=False,want_covs=True)[1]
    S=sp.PointCloudShared(dev(src)); S.covs=sp.GridKNN.build(S.points,points_per_cell=6.0).self_knn(20,want_knn=False,want_covs=True)[1]
    S=S.reordered(sp.GridKNN.build(S.points,points_per_cell=1.0).order())
    prep=sp.PreparedTarget(sp.GridKNN.build(Tg.points,points_per_cell=0.5),Tg.covs)
    p=sp.RegistrationParams(criteria_translation=0.0,criteria_rotation=0.0,max_iterations=12)
    ident=torch.eye(4,device="cuda").reshape(-1).contiguous()
    def run(**kw):
        reg=sp.Registration(p); T_dev=ident.clone(); delta=torch.zeros(8,device="cuda"); out=[]
        for _ in range(6):
            T_dev.copy_(ident)
            reg.align_fused_loop(S,prep,iterations=12,T_dev=T_dev,delta_dev=delta,sort_by_cell="presorted",**kw)
            torch.cuda.synchronize(); out.append(T_dev.cpu().numpy().copy())
        return out
    ref=run()
    for name,kw in (("rccl-graph",dict(comm=comm,graph=True)),("rccl-graph2",dict(comm=comm,graph=True))):
        o=run(**kw)
        print(n,name,[float(np.abs(x-ref[0]).max()) for x in o], flush=True)
