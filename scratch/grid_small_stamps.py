"""Phase stamps of grid_build_small_kernel (development build with SB_STAMP compiled in) on the example's downsampled target."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sycl_points_amd.api as sp
from sycl_points_amd import _lib

def read(path):
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    n = int([l for l in head.split(b"\n") if l.startswith(b"element vertex")][0].split()[-1])
    a = np.frombuffer(body, dtype="<f4", count=n * 4).reshape(n, 4)
    p = np.ones((n, 4), np.float32); p[:, :3] = a[:, :3]
    return p
pts = read(os.path.join(ROOT, "tests/golden/target.ply"))
linf = np.abs(pts[:, :3]).max(1)
cloud = torch.from_numpy(pts[(linf >= 0.5) & (linf <= 50)]).cuda()
down = sp.VoxelGrid(0.25).downsampling(cloud).points.contiguous()
print("points", down.shape[0])
L = _lib.lib()
L.sp_internal_grid_small_stamps.restype = C.c_int
L.sp_internal_grid_small_stamps.argtypes = [C.c_void_p]
for rep in range(4):
    g = sp.GridKNN.build(down, points_per_cell=0.5)
    torch.cuda.synchronize()
    st = (C.c_ulonglong * 16)()
    L.sp_internal_grid_small_stamps(C.cast(st, C.c_void_p))
    s = [int(x) for x in st]
    names = {0: "start", 1: "keys", 2: "pass0", 3: "pass1", 4: "pass2", 6: "sorted", 7: "gathered", 8: "scan", 9: "end"}
    seq = [(i, s[i]) for i in (0, 1, 2, 3, 4, 6, 7, 8, 9) if s[i] >= s[0] and s[i] != 0]
    print(" ".join(f"{names[i]}+{(t - s[0]) / 100:.2f}us" for i, t in seq))
