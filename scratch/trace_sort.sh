#!/bin/bash
# kernel durations of the voxel / grid-build stages (bench_stages.py --only voxel) under rocprofv3 --stats; GPU box, repo root
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_rs
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rs -- python3 $GRAFT_REPO_ROOT/bench_stages.py --only voxel > $GRAFT_REPO_ROOT/gpurun_out/rs/stages.json 2> $GRAFT_REPO_ROOT/gpurun_out/rs/stages.err
cp $(find /tmp/prof_rs -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/rs/kernel_stats.csv
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, json
for r in csv.DictReader(open("gpurun_out/rs/kernel_stats.csv")):
    if any(k in r["Name"] for k in ("rs_", "aggregate", "scatter_kernel", "key32", "fillBuffer", "zero_")):
        print(r["Name"][:80], r["Calls"], round(float(r["AverageNs"])/1e3, 1))
d = json.load(open("gpurun_out/rs/stages.json"))
print({k: (v if not isinstance(v, dict) else {kk: round(vv, 4) for kk, vv in v.items() if "ms" in kk}) for k, v in d.items() if "voxel" in k or "grid_build" in k})
PY
