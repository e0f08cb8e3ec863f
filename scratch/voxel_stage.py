"""bench.py's two voxel stages alone (1 M points, voxel 0.1, R = 10 / 2.5; sp_voxel_downsample_report with the key box known), for
A/B runs of two libraries on one box; also the example's box-filtered target scan (63 985 points, 0.25 m)."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sycl_points_amd.api as sp
from sycl_points_amd import _lib
from sycl_points_amd.synthetic import Mt19937Cloud

L = _lib.lib()
def stage(P, vs):
    n = P.shape[0]
    vg = sp.VoxelGrid(vs)
    nvox = vg.downsampling(P).size()
    box = vg._key_box
    nbytes = L.sp_voxel_downsample_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=P.device)
    o_p = torch.empty((n, 4), dtype=torch.float32, device=P.device)
    info = torch.zeros(8, dtype=torch.int32, device=P.device)
    def run():
        _lib.check(L.sp_voxel_downsample_report(sp._ptr(P), n, vg.voxel_size_inv, 1, None, None, None, sp._ptr(o_p), None, None, None, None,
                                                None, box.ctypes.data_as(C.c_void_p), sp._ptr(info), sp._ptr(ws), nbytes, sp._stream()))
    ms = []
    for _ in range(31):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    assert int(info[0]) == nvox
    return float(np.median(ms[5:]))
out = []
for R in (10.0, 2.5):
    out.append(stage(torch.from_numpy(Mt19937Cloud(1234).uniform_points(1_000_000, R)).cuda(), 0.1))
def read(path):
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    n = int([l for l in head.split(b"\n") if l.startswith(b"element vertex")][0].split()[-1])
    a = np.frombuffer(body, dtype="<f4", count=n * 4).reshape(n, 4)
    p = np.ones((n, 4), np.float32); p[:, :3] = a[:, :3]
    return p
pts = read(os.path.join(ROOT, "tests/golden/target.ply"))
linf = np.abs(pts[:, :3]).max(1)
out.append(stage(torch.from_numpy(pts[(linf >= 0.5) & (linf <= 50)]).cuda(), 0.25))
print("voxel 1M sparse %.4f ms, dense %.4f ms, scan 64k %.4f ms" % tuple(out))
