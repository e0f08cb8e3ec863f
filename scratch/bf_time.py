import torch, numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud
g = Mt19937Cloud(1234)
t = torch.from_numpy(g.uniform_points(100000, 10.0)).cuda(); q = torch.from_numpy(g.uniform_points(100000, 10.0)).cuda()
ks = [int(a) for a in sys.argv[1:]] or [1, 5, 10, 20]
for k in ks:
    sp.knn_search_bruteforce(q, t, k); torch.cuda.synchronize()
    ts = []
    for _ in range(11):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); sp.knn_search_bruteforce(q, t, k); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(k, np.median(ts))
