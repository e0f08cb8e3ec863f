"""How long is the one-lane Gauss-Newton step (6x6 pivoted LDL^T + se3_exp + pose update)? sp_gn_update in a loop against an
empty-ish kernel, HIP events over 2000 back-to-back launches."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_points_amd.api as sp
from sycl_points_amd import _lib
L = _lib.lib()
lin = torch.zeros(48, dtype=torch.float32, device="cuda")
H = np.eye(6, dtype=np.float32) * 5 + 0.1
lin[:36] = torch.from_numpy(H.reshape(-1)).cuda(); lin[36:42] = 0.01
T = torch.eye(4, dtype=torch.float32, device="cuda").reshape(-1).contiguous()
delta = torch.zeros(8, dtype=torch.float32, device="cuda")
z = torch.zeros(1, device="cuda")
def loop(fn, n=2000):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
st = sp._stream()
t_solve = loop(lambda: L.sp_gn_update(sp._ptr(lin), sp._ptr(T), C.c_float(1.0), C.c_float(0.0), C.c_float(0.0), sp._ptr(delta), st))
t_zero = loop(lambda: L.sp_cov_update_plane(sp._ptr(lin), 1, sp._ptr(lin), st))
print(f"sp_gn_update {t_solve:.2f} us per launch; a one-thread trivial kernel {t_zero:.2f} us")
