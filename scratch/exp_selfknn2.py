import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud
pts=Mt19937Cloud(1234).uniform_points(1000000,10.0); P=torch.from_numpy(pts).cuda()
for ppc,k in ((16.0,20),(8.0,20),(6.0,20)):
    g=sp.GridKNN.build(P,points_per_cell=ppc)
    for _ in range(2): g.self_knn(k,True,False,False)
    torch.cuda.synchronize()
    print("ppc",ppc,"k",k,"todo count:", int(g._keep[:4].view(torch.int32)[0]), flush=True)
