import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud
g=Mt19937Cloud(1234)
pts=torch.from_numpy(g.uniform_points(1000000,10.0)).cuda()
for ppc in (6.0, 0.5):
    for _ in range(3): sp.GridKNN.build(pts,points_per_cell=ppc)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): gr=sp.GridKNN.build(pts,points_per_cell=ppc)
    torch.cuda.synchronize(); print("grid build ppc",ppc,"ms",(time.perf_counter()-t)/10*1e3)
