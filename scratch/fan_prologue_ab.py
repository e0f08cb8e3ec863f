"""the single-GPU launches of config 4 with the prologue reading ONE row of totals (summed by the previous launch's
last-arriving workgroup) against every workgroup summing the 256 partial rows itself — same box, interleaved, bit-compared.
python scratch/fan_prologue_ab.py   (GPU box)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sycl_points_amd.api as sp
from sycl_points_amd.synthetic import gicp_pair

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
n = 1_000_000
src, tgt, T_gt = gicp_pair(n, 10.0)
Tg = sp.PointCloudShared(dev(tgt))
Tg.covs = sp.GridKNN.build(Tg.points, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
grid = sp.GridKNN.build(Tg.points, points_per_cell=0.5)
S_all = dev(src)
S_all = S_all[sp.GridKNN.build(S_all, points_per_cell=1.0).order()].contiguous()
covs = sp.GridKNN.build(S_all, points_per_cell=6.0).self_knn(20, want_knn=False, want_covs=True)[1]
S = sp.PointCloudShared(S_all, covs=covs)
prep = sp.PreparedTarget(grid, Tg.covs)
p = sp.RegistrationParams(max_iterations=20, criteria_rotation=0.0, criteria_translation=0.0)
regs = {}
out = {}
for fan in (0, 1):
    reg = sp.Registration(p)
    reg._set_source_option("fan_prologue", fan)
    regs[fan] = reg
    T_dev, lin, delta = reg.align_fused_loop(S, prep, sort_by_cell="presorted")
    torch.cuda.synchronize()
    out[fan] = (T_dev.cpu().numpy().copy(), lin.cpu().numpy().copy(), delta.cpu().numpy().copy(), int(reg._iters_dev[0]))
same = all(np.array_equal(a, b) for a, b in zip(out[0][:3], out[1][:3])) and out[0][3] == out[1][3]
print("bit-identical:", same, "iterations", out[0][3], out[1][3])
ms = {0: [], 1: []}
for rep in range(12):
    for fan in (0, 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            regs[fan].align_fused_loop(S, prep, sort_by_cell="presorted")
        e1.record(); torch.cuda.synchronize()
        ms[fan].append(e0.elapsed_time(e1) / 5)
for fan in (0, 1):
    m = np.median(ms[fan][2:])
    print(f"fan_prologue={fan}: {m:.4f} ms per alignment, {m / out[fan][3] * 1e3:.2f} us per iteration (min {min(ms[fan]):.4f})")
