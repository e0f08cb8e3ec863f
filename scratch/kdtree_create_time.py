import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import Mt19937Cloud
L = sp._lib.lib()
torch.zeros(1, device="cuda")
for n in (6096, 69088, 1000000):
    P = Mt19937Cloud(7).uniform_points(n, 10.0)
    ts = []
    for _ in range(12):
        h = C.c_void_p()
        t0 = time.perf_counter()
        sp.check(L.sp_kdtree_create(P.ctypes.data_as(C.c_void_p), n, 16, sp._stream(), C.byref(h)))
        t1 = time.perf_counter()
        L.sp_kdtree_destroy(h)
        ts.append((t1 - t0) * 1e3)
    print(n, "sp_kdtree_create ms: median %.3f min %.3f" % (np.median(ts[2:]), min(ts[2:])))
