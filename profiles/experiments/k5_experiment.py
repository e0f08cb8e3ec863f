"""K5 (covariance from kNN lists, 1 M points, k = 20): shipped kernel vs the MFMA experiment, on the cloud as generated
(random order) and on the same cloud in cell order (what voxel downsampling / GridKNN.order() yields).
Run on the GPU box from the repo root: python profiles/experiments/k5_experiment.py > gpurun_out/k5_experiment.txt"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import sycl_points_amd.api as sp
from sycl_points_amd import _lib
from sycl_points_amd.synthetic import Mt19937Cloud

X = C.CDLL(os.path.join(os.path.dirname(__file__), "libcov_mfma.so"))
X.exp_cov_mfma.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
L = _lib.lib()


def median(fn, runs=21):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(runs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


n, k = 1_000_000, 20
P0 = torch.from_numpy(Mt19937Cloud(1234).uniform_points(n, 10.0)).cuda()
for name, P in (("random order", P0), ("cell order", None)):
    if P is None:
        P = P0[sp.GridKNN.build(P0, points_per_cell=1.0).order()].contiguous()
    grid = sp.GridKNN.build(P, points_per_cell=6.0)
    res = grid.self_knn(k, True, False, False)[0]
    idx = res.indices
    covs = torch.empty((n, 4, 4), dtype=torch.float32, device="cuda")
    covs_m = torch.empty_like(covs)
    st = sp._stream()
    t_valu = median(lambda: _lib.check(L.sp_cov_estimate(sp._ptr(P), n, sp._ptr(idx), k, sp._ptr(covs), st)))
    t_mfma = median(lambda: X.exp_cov_mfma(sp._ptr(P), n, sp._ptr(idx), k, sp._ptr(covs_m), st))
    d = (covs - covs_m).abs()
    differing = int((d.reshape(n, -1).max(dim=1).values > 0).sum())
    print(f"{name}: shipped K5 {t_valu:.4f} ms = {464e6 / t_valu / 1e6:.0f} GB/s algorithmic = {464e6 / t_valu / 1e6 / 8000:.3f} of HBM; "
          f"MFMA 4x4x1 form {t_mfma:.4f} ms; covariances that differ from the bit-exact kernel: {differing} of {n} "
          f"(max abs diff {float(d.max()):.3g}, max |cov| {float(covs.abs().max()):.3g})", flush=True)
