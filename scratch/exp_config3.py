import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud
def timed(fn,reps=10):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps*1e3
for R in (10.0,2.5):
    pts=Mt19937Cloud(1234).uniform_points(1000000,R); P=torch.from_numpy(pts).cuda()
    vg=sp.VoxelGrid(0.1)
    out=vg.downsampling(P)
    print("R=%.1f: voxel downsample 1M -> %d voxels: %.0f us ; keys only %.1f us"%(R,out.size(),timed(lambda: vg.downsampling(P)),timed(lambda: vg.compute_voxel_bit(P))))
