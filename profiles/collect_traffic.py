#!/usr/bin/env python3
"""Collect HBM traffic of the hot kernels with rocprofv3 PMC counters and write profiles/traffic.json.

Run on the GPU box from the repository root:  python3 profiles/collect_traffic.py
Method (MI355X_MICROARCH.md §HBM / cdna_hip_programming.md §7): FETCH_SIZE and WRITE_SIZE are collected in SEPARATE
passes (they do not fit one pass on gfx950), with --kernel-trace only; both are in KiB. The guide's gfx950 correction
(FETCH_SIZE reads exactly half for WIDE COALESCED streaming reads) does not transfer to gather-shaped kernels, so, as the
guide prescribes for other access shapes, the read side is CALIBRATED on a known byte count in the same access pattern:
prepare_cov_kernel gathers one 64-byte covariance row per lane through an index (the shape of the fused kernel's loads)
and reads a known n x (64 + 16) bytes. factor = known / FETCH_SIZE; WRITE_SIZE is exact for 16-byte stores (checked:
prepare_cov writes exactly n x 32 bytes). bytes = factor * FETCH_SIZE * 1024 + WRITE_SIZE * 1024, per launch. This script never touches the GPU itself; rocprofv3 wraps `python3 bench.py` directly.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "pmc_traffic")
KERNELS = {"gicp_align_kernel": "gicp_align_kernel", "align_finish_kernel": "align_finish_kernel",
           "gicp_fused_kernel": "gicp_fused_kernel", "final_reduce_kernel": "final_reduce_kernel",
           "prepare_cov_kernel": "prepare_cov_kernel"}


def run_pass(counter):
    d = os.path.join(OUT, counter)
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "40", "--warmup", "20"]
    subprocess.run(cmd, check=True, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return read_pass(counter)


def read_pass(counter):
    d = os.path.join(OUT, counter)
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        for key, pat in KERNELS.items():
            if pat in r["Kernel_Name"]:
                acc[key].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    if "--reuse" in sys.argv:  # recompute from CSVs already under gpurun_out/pmc_traffic (no GPU needed)
        fetch, write = read_pass("FETCH_SIZE"), read_pass("WRITE_SIZE")
    else:
        fetch = run_pass("FETCH_SIZE")
        write = run_pass("WRITE_SIZE")
    out = {}
    n = 1_000_000
    known_prepare_read = n * (64 + 16)  # prepare_cov_kernel (target side): one 64-B covariance row gathered through the index in a 16-B grid point
    factor = known_prepare_read / (fetch["prepare_cov_kernel"][0] * 1024.0)
    for k in KERNELS:
        if k in fetch and k in write:
            f_kib, nf = fetch[k]
            w_kib, _ = write[k]
            out[k] = {"FETCH_SIZE_KiB_per_launch": f_kib, "WRITE_SIZE_KiB_per_launch": w_kib, "launches": nf,
                      "read_calibration_factor": factor,
                      "hbm_bytes_per_launch": factor * f_kib * 1024.0 + w_kib * 1024.0,
                      "uncorrected_bytes_per_launch": f_kib * 1024.0 + w_kib * 1024.0,
                      "correction": "read side calibrated on prepare_cov_kernel (known n*(64+16) B, same gather shape); "
                                    "WRITE_SIZE exact"}
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
