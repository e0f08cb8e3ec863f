#!/usr/bin/env python3
"""Per-stage throughputs of BASELINE configs 2 and 3 (and the pre-loop stages of config 4) on one MI355X.

Not the driver's contract benchmark (that is bench.py): this prints one JSON object with the stage timings quoted in
DESIGN.md §4: brute-force KNN (100k x 100k, k=1 and k=20), voxel keys / voxel-grid downsampling (1M points, voxel 0.1,
R=10 sparse and R=2.5 dense), KNN k=20 (KD-tree, GridKNN) and covariance (K5 alone, fused self-kNN + covariance).
Each stage: inputs resident in HBM, HIP events on the launch stream, mean of `reps` launches after one warm-up.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sycl_points_amd.api as sp  # noqa: E402
from sycl_points_amd.synthetic import Mt19937Cloud  # noqa: E402

HBM = 8000.0
FP32 = 157.3e12


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3  # seconds


def main():
    out = {}
    g = Mt19937Cloud(1234)
    tgt = torch.from_numpy(g.uniform_points(100000, 10.0)).cuda()
    qry = torch.from_numpy(g.uniform_points(100000, 10.0)).cuda()
    for k in (1, 20):
        t = timed(lambda: sp.knn_search_bruteforce(qry, tgt, k))
        pairs = 1e10
        out[f"bruteforce_100k_k{k}"] = {"ms": t * 1e3, "queries_per_s": 1e5 / t, "pairs_per_s": pairs / t,
                                        "valu_frac_of_fp32_peak_at_9_ops_per_pair": 9 * 2 * pairs / t / 2 / FP32,
                                        "hbm_GBps_algorithmic": (16 * 2e5 + 8 * 1e5 * k) / t / 1e9}
    for name, R in (("sparse_R10", 10.0), ("dense_R2.5", 2.5)):
        P = torch.from_numpy(Mt19937Cloud(1234).uniform_points(1000000, R)).cuda()
        vg = sp.VoxelGrid(0.1)
        t_keys = timed(lambda: vg.compute_voxel_bit(P))
        nvox = vg.downsampling(P).size()
        t_ds = timed(lambda: vg.downsampling(P))
        out[f"voxel_1M_{name}"] = {"voxels": nvox, "keys_ms": t_keys * 1e3, "keys_GBps": 24e6 / t_keys / 1e9,
                                   "keys_frac_hbm": 24e6 / t_keys / 1e9 / HBM, "downsample_ms": t_ds * 1e3,
                                   "downsample_points_per_s": 1e6 / t_ds, "downsample_GBps_at_40B": 40e6 / t_ds / 1e9}
    pts = Mt19937Cloud(1234).uniform_points(1000000, 10.0)
    P = torch.from_numpy(pts).cuda()
    tree = sp.KDTree.build(pts)
    r = sp.KNNResult()
    t_kd = timed(lambda: tree.knn_search_async(P, 20, r), 3)
    t_k5 = timed(lambda: sp.covariance.estimate(r, P))
    grid = sp.GridKNN.build(P, points_per_cell=6.0)
    t_gk = timed(lambda: grid.self_knn(20, True, False, False), 3)
    t_gc = timed(lambda: grid.self_knn(20, False, True, False), 3)
    out["knn20_1M"] = {"kdtree_ms": t_kd * 1e3, "kdtree_GBps_at_176B": 176e6 / t_kd / 1e9,
                       "grid_self_ms": t_gk * 1e3, "grid_self_GBps_at_176B": 176e6 / t_gk / 1e9}
    out["covariance_1M_k20"] = {"K5_ms": t_k5 * 1e3, "K5_GBps_at_464B": 464e6 / t_k5 / 1e9, "K5_frac_hbm": 464e6 / t_k5 / 1e9 / HBM,
                                "kdtree_knn_plus_K5_ms": (t_kd + t_k5) * 1e3, "grid_fused_knn_cov_ms": t_gc * 1e3,
                                "grid_fused_points_per_s": 1e6 / t_gc}
    import time
    t0 = time.time(); sp.KDTree.build(pts); out["kdtree_build_1M_host_ms"] = (time.time() - t0) * 1e3
    torch.cuda.synchronize(); t0 = time.time(); sp.GridKNN.build(P, points_per_cell=0.5); torch.cuda.synchronize()
    out["grid_build_1M_device_ms"] = (time.time() - t0) * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
