"""Per-frame pipeline timing (host wall clock, synchronised): what a caller pays for one new target + one new source."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import gicp_pair
n=1000000
src,tgt,T=gicp_pair(n,10.0)
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
tp,spn=dev(tgt),dev(src)
def wall(fn,reps=5):
    fn(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(reps): r=fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/reps*1e3
vg=sp.VoxelGrid(0.1)
print("voxel downsample      %.3f ms"%wall(lambda: vg.downsampling(sp.PointCloudShared(tp))))
print("grid build (ppc 6)    %.3f ms"%wall(lambda: sp.GridKNN.build(tp,points_per_cell=6.0)))
g6=sp.GridKNN.build(tp,points_per_cell=6.0)
print("self kNN20 + cov      %.3f ms"%wall(lambda: g6.self_knn(20,want_knn=False,want_covs=True)))
covs=g6.self_knn(20,want_knn=False,want_covs=True)[1]
print("grid build (ppc 0.5)  %.3f ms"%wall(lambda: sp.GridKNN.build(tp,points_per_cell=0.5)))
g=sp.GridKNN.build(tp,points_per_cell=0.5)
print("prepared target       %.3f ms"%wall(lambda: sp.PreparedTarget(g,covs)))
prep=sp.PreparedTarget(g,covs)
gs=sp.GridKNN.build(spn,points_per_cell=6.0)
scov=gs.self_knn(20,want_knn=False,want_covs=True)[1]
order=sp.GridKNN.build(spn,points_per_cell=1.0).order()
S=sp.PointCloudShared(spn[order].contiguous(),covs=scov[order].contiguous())
p=sp.RegistrationParams(max_iterations=20,criteria_translation=0.0,criteria_rotation=0.0)
reg=sp.Registration(p)
def align():
    return reg.align_fused_loop(S,prep,iterations=20,sort_by_cell="presorted")
print("align 20 iterations   %.3f ms"%wall(align))
print("grid order of source  %.3f ms"%wall(lambda: sp.GridKNN.build(spn,points_per_cell=1.0).order()))
print("self kNN k=3 on the ppc 0.5 grid  %.3f ms"%wall(lambda: g.self_knn(3,want_knn=True,want_covs=False)))
print("self kNN k=3 on the ppc 6 grid    %.3f ms"%wall(lambda: g6.self_knn(3,want_knn=True,want_covs=False)))
g2=sp.GridKNN.build(tp,points_per_cell=2.0)
print("self kNN k=3 on a ppc 2 grid      %.3f ms"%wall(lambda: g2.self_knn(3,want_knn=True,want_covs=False)))
from sycl_points_amd import _lib
[x._set_option('self_knn_mode',1) for x in (g,g2,g6)]
print("mode 1 (lane per query): k=3 ppc 0.5 %.3f ms | ppc 2 %.3f ms | ppc 6 %.3f ms"%(wall(lambda: g.self_knn(3,want_knn=True,want_covs=False)),wall(lambda: g2.self_knn(3,want_knn=True,want_covs=False)),wall(lambda: g6.self_knn(3,want_knn=True,want_covs=False))))
a=g.self_knn(3,want_knn=True,want_covs=False)[0]
[x._set_option('self_knn_mode',0) for x in (g,g2,g6)]
b=g.self_knn(3,want_knn=True,want_covs=False)[0]
print("same neighbours:", bool((a.indices==b.indices).all()), bool((a.distances==b.distances).all()))
