#!/bin/bash
# A/B of a scratch library against the shipped one on one box: per-launch profile of bench.py.
# usage (GPU box, repo root): bash scratch/ab_lib.sh scratch/lib_exp_0.so [extra bench flags]
cd $GRAFT_REPO_ROOT
LIB=$1; shift
mkdir -p gpurun_out/ab
cp sycl_points_amd/lib/libsycl_points_amd.so /tmp/lib_keep.so
for tag in A B A B; do
  if [ $tag = A ]; then cp /tmp/lib_keep.so sycl_points_amd/lib/libsycl_points_amd.so; else cp $LIB sycl_points_amd/lib/libsycl_points_amd.so; fi
  timeout -k 10 150 python bench.py --repeats 5 --no-cpu-baseline "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err || { echo "$tag failed"; tail -3 gpurun_out/ab/$tag.err; continue; }
  python - gpurun_out/ab/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
l = d["launches_of_one_alignment"]
print(sys.argv[2], "us", [x["us"] for x in l[:8]], "us/step", round(d["ms_per_step"] * 1e3, 2), "conv ms", round(d["until_converged"]["ms_per_alignment"], 4),
      "pose err", d["pose_max_abs_err_vs_ground_truth"])
PY
done
cp /tmp/lib_keep.so sycl_points_amd/lib/libsycl_points_amd.so
