"""Phase stamps of aggregate_kernel (development build) on the example's box-filtered target scan (63 985 points, 0.25 m voxels)."""
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
L = _lib.lib()
L.sp_internal_ag_stamps.restype = C.c_int
L.sp_internal_ag_stamps.argtypes = [C.c_void_p, C.c_int]
vg = sp.VoxelGrid(0.25)
vg.downsampling(cloud)
for rep in range(4):
    L.sp_internal_ag_stamps(None, 1)
    vg.downsampling(cloud)
    torch.cuda.synchronize()
    st = (C.c_ulonglong * 16)()
    L.sp_internal_ag_stamps(C.cast(st, C.c_void_p), 0)
    s = [int(x) for x in st]
    n = max(s[8], 1)
    names = ["loads + scan in the wave", "across waves + heads", "runs past the tile", "outputs"]
    print(f"workgroups {s[8]}, first start to last end {(s[10] - s[9]) / 100:.2f} us; " +
          "; ".join(f"{names[j]} max {s[j] / 100:.2f} mean {s[4 + j] / n / 100:.2f}" for j in range(4)))
