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
           sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "40", "--warmup", "20", "--repeats", "3"]
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


GATHER_FACTOR = 1.13  # read-side calibration for gather-shaped launches, measured in round 1 (r01_g) on
                      # prepare_cov_kernel (known n x 80 B gathered -> FETCH_SIZE x 1.13); since the correspondence cache
                      # was added that kernel's window also sees write-backs of older dirty lines, so the constant is kept


def per_launch(counter, pattern):
    d = os.path.join(OUT, counter)
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    rows = [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter and pattern in r["Kernel_Name"]]
    return [v for _, v in sorted(rows)]


def main():
    if "--reuse" not in sys.argv:  # otherwise recompute from CSVs already under gpurun_out/pmc_traffic (no GPU needed)
        run_pass("FETCH_SIZE")
        run_pass("WRITE_SIZE")
    out = {}
    for key, pat in KERNELS.items():
        f, w = per_launch("FETCH_SIZE", pat), per_launch("WRITE_SIZE", pat)
        if not f or len(f) != len(w):
            continue
        if key == "gicp_align_kernel":
            # Steady-state launches (every correspondence certified from the source-ordered cache, nothing written back
            # to it) are wide coalesced streams: the guide's gfx950 rule applies, FETCH_SIZE reads half the bytes (check:
            # 2 x 47.4 MiB = 99.4 MB against the 96 B/point the kernel is known to stream). Launches that search (the first
            # poses of an alignment: cache rows rewritten, WRITE_SIZE in the tens of MiB) are gather-shaped: calibrated
            # factor. Both in KiB as the counters report.
            # (a searched point rewrites its 48-byte cache row: 480 KiB of writes = 1 % of 1M points searched, the same
            # threshold bench.py's launch_classes uses on the device-side searched-point count)
            # launches that return at once (an earlier iteration converged: bench.py's until_converged leg) fetch < 1 MiB
            # and belong to neither class
            steady = [(2.0 * a + b) * 1024.0 for a, b in zip(f, w) if b < 480.0 and a >= 1024.0]
            search = [(GATHER_FACTOR * a + b) * 1024.0 for a, b in zip(f, w) if b >= 480.0]
            allb = steady + search
            out[key] = {"launches": len(allb), "hbm_bytes_per_launch": sum(allb) / len(allb),
                        "steady_state_launches": len(steady),
                        "steady_state_hbm_bytes_per_launch": sum(steady) / max(len(steady), 1),
                        "searching_launches": len(search),
                        "searching_hbm_bytes_per_launch": sum(search) / max(len(search), 1),
                        "classes": {"steady": {"launches": len(steady),
                                               "hbm_bytes_per_launch": sum(steady) / max(len(steady), 1)},
                                    "searching": {"launches": len(search),
                                                  "hbm_bytes_per_launch": sum(search) / max(len(search), 1)}},
                        "FETCH_SIZE_KiB_per_launch": sum(f) / len(f), "WRITE_SIZE_KiB_per_launch": sum(w) / len(w),
                        "correction": "steady-state launches: FETCH_SIZE x 2 (wide coalesced stream, guide's gfx950 rule); "
                                      "searching launches: FETCH_SIZE x 1.13 (gather-shaped, calibrated); WRITE_SIZE exact"}
        else:
            fm, wm = sum(f) / len(f), sum(w) / len(w)
            out[key] = {"launches": len(f), "FETCH_SIZE_KiB_per_launch": fm, "WRITE_SIZE_KiB_per_launch": wm,
                        "hbm_bytes_per_launch": (GATHER_FACTOR * fm + wm) * 1024.0,
                        "correction": "FETCH_SIZE x 1.13 (gather-shaped, calibrated); WRITE_SIZE exact"}
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
