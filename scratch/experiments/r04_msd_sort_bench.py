"""bucket sort (msd_sort_pairs_u32) against the stable radix passes on voxel-like keys: same output? time?"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sycl_points_amd.api as sp
from sycl_points_amd import _lib
L = _lib.lib()
P = lambda t: C.c_void_p(t.data_ptr())


def med(fn, runs=11):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(runs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1))
    return float(np.median(t))


n = 1_000_000
rs = np.random.RandomState(1)
cases = {"sparse 200^3 (23 bits)": (rs.randint(0, 200 ** 3, n), 23), "dense 50^3 (17 bits)": (rs.randint(0, 50 ** 3, n), 17),
         "clustered (23 bits, half the keys in 3 buckets)": (np.where(rs.rand(n) < 0.5, rs.randint(0, 6000, n), rs.randint(0, 200 ** 3, n)), 23)}
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, (k, bits) in cases.items():
    keys = torch.from_numpy(k.astype(np.uint32).view(np.int32)).cuda()
    # LSD reference
    ka, kb = keys.clone(), torch.empty_like(keys)
    va, vb = torch.arange(n, dtype=torch.int32, device="cuda"), torch.empty(n, dtype=torch.int32, device="cuda")
    wsb = L.sp_internal_radix_sort_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    in_b = C.c_int(0)

    def lsd():
        ka.copy_(keys); va.copy_(torch.arange(n, dtype=torch.int32, device="cuda"))
        _lib.check(L.sp_internal_radix_sort_u32(P(ka), P(kb), P(va), P(vb), n, bits, P(ws), wsb, C.byref(in_b), st))
    lsd(); torch.cuda.synchronize()
    rk, rv = (kb, vb) if in_b.value else (ka, va)
    rk, rv = rk.clone(), rv.clone()
    t_copy = med(lambda: (ka.copy_(keys), va.copy_(torch.arange(n, dtype=torch.int32, device="cuda"))))
    t_lsd = med(lsd) - t_copy
    mwsb = L.sp_internal_msd_sort_workspace_bytes(n)
    mws = torch.empty(mwsb, dtype=torch.uint8, device="cuda")
    ok_, ov_ = torch.empty_like(keys), torch.empty_like(keys)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")

    def msd():
        flag.zero_()
        _lib.check(L.sp_internal_msd_sort_u32(P(keys), n, bits, P(ok_), P(ov_), P(flag), P(mws), mwsb, st))
    msd(); torch.cuda.synchronize()
    over = int(flag[0]) != 0
    same = (not over) and torch.equal(ok_, rk) and torch.equal(ov_, rv)
    t_msd = med(msd)
    print(f"{name}: radix passes {t_lsd * 1e3:.1f} us, bucket sort {t_msd * 1e3:.1f} us, overflow {over}, identical {same}", flush=True)
