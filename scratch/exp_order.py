import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import gicp_pair
def morton(p, bits=10):
    mn=p[:,:3].min(0); mx=p[:,:3].max(0)
    q=((p[:,:3]-mn)/(mx-mn+1e-9)*(2**bits-1)).astype(np.uint64)
    def spread(v):
        v=v&0x3FF; v=(v|(v<<16))&0x30000FF; v=(v|(v<<8))&0x300F00F; v=(v|(v<<4))&0x30C30C3; v=(v|(v<<2))&0x9249249; return v
    return spread(q[:,0])|(spread(q[:,1])<<1)|(spread(q[:,2])<<2)
def timed(fn,reps=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps*1e3
n=1000000
src,tgt,T=gicp_pair(n,10.0)
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for tname,tg in (("tgt random",tgt),("tgt morton",tgt[np.argsort(morton(tgt))])):
    Tg=sp.PointCloudShared(dev(tg)); tree=sp.KDTree.build(tg)
    sp.covariance.estimate(tree.knn_search(Tg,20),Tg)
    for sname,sr in (("src random",src),("src morton",src[np.argsort(morton(src))])):
        S=sp.PointCloudShared(dev(sr)); st=sp.KDTree.build(sr); sp.covariance.estimate(st.knn_search(S,20),S)
        reg=sp.Registration(sp.RegistrationParams(criteria_translation=0.0,criteria_rotation=0.0))
        Td=dev(np.eye(4,dtype=np.float32).reshape(-1))
        t_nn0=timed(lambda: tree.nearest_neighbor_search_async(S,reg.neighbors,Td))
        reg.align_device_loop(S,Tg,tree,iterations=20,T_dev=Td)
        t_nn=timed(lambda: tree.nearest_neighbor_search_async(S,reg.neighbors,Td))
        t_k11=timed(lambda: reg._linearize("linearize",S,Tg,Td,10.0,reg._lin))
        r20=sp.KNNResult(); t_k20=timed(lambda: tree.knn_search_async(S,20,r20),reps=3)
        print(f"{tname:12s} {sname:12s} NN(identity) {t_nn0:7.1f} us  NN(converged) {t_nn:7.1f} us  K11 {t_k11:6.1f} us  KNN20 {t_k20:8.1f} us", flush=True)
