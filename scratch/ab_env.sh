#!/bin/bash
# A/B/... of scratch libraries on one box through SP_AMD_LIB (the shipped library is not touched), two rounds, alternating:
#   bash scratch/ab_env.sh scratch/exp_libs/lib_0.so scratch/exp_libs/lib_1.so ...    (BENCH_FLAGS="..." for extra bench flags)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for round in 1 2; do
for L in "$@"; do
  tag=$(basename $L .so)
  SP_AMD_LIB=$GRAFT_REPO_ROOT/$L timeout -k 10 200 python bench.py --repeats 9 --no-cpu-baseline --no-stages $BENCH_FLAGS > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err || { echo "$tag failed"; tail -3 gpurun_out/ab/$tag.err; exit 1; }
  python - gpurun_out/ab/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
l = d["launches_of_one_alignment"]
print(sys.argv[2], "us", [x["us"] for x in l[:6]], "us/step", round(d["ms_per_step"] * 1e3, 2), "conv ms", round(d["until_converged"]["ms_per_alignment"], 4),
      "pose err", d["pose_max_abs_err_vs_ground_truth"], "inl", d["inliers_last_iteration"])
PY
done
done
