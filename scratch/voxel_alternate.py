"""Two scans through ONE VoxelGrid alternately (the reference's example does this): does the remembered key box of one
scan cover the other, or is every call redone?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sycl_points_amd.api as sp
def ply(path):
    raw = open(path, "rb").read(); head, body = raw.split(b"end_header\n", 1)
    n = int([l for l in head.split(b"\n") if l.startswith(b"element vertex")][0].split()[-1])
    a = np.ones((n, 4), np.float32); a[:, :3] = np.frombuffer(body, dtype="<f4", count=n * 4).reshape(n, 4)[:, :3]; return a
S = torch.from_numpy(ply(os.path.join(ROOT, "tests/golden/source.ply"))).cuda()
T = torch.from_numpy(ply(os.path.join(ROOT, "tests/golden/target.ply"))).cuda()
vg = sp.VoxelGrid(0.25)
orig = sp._lib.lib().sp_voxel_downsample_boxed
calls = [0]
class W:
    def __call__(self, *a):
        calls[0] += 1
        return orig(*a)
sp._lib.lib().sp_voxel_downsample_boxed = W()
for i in range(6):
    c0 = calls[0]; t0 = time.perf_counter()
    out = vg.downsampling(S if i % 2 == 0 else T)
    torch.cuda.synchronize()
    print(i, "boxed calls", calls[0] - c0, "voxels", out.size(), "ms %.3f" % ((time.perf_counter() - t0) * 1e3), "box", vg._key_box.tolist())
