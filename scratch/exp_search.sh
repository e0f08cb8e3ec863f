#!/bin/bash
# Where a searching launch's time goes: the measurement builds of the per-iteration kernel (make -C sycl_points_amd/csrc exp EXP=n,
# results wrong by construction) against the shipped library with every launch searching every point (--internal reuse=0).
# usage (GPU box, repo root): bash scratch/exp_search.sh [extra bench flags]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp_search
cp sycl_points_amd/lib/libsycl_points_amd.so /tmp/lib_keep.so
for v in base 1 2 4 7; do
  if [ $v = base ]; then cp /tmp/lib_keep.so sycl_points_amd/lib/libsycl_points_amd.so; else cp scratch/lib_exp_$v.so sycl_points_amd/lib/libsycl_points_amd.so; fi
  timeout -k 10 150 python bench.py --internal reuse=0 --repeats 3 --no-cpu-baseline "$@" > gpurun_out/exp_search/v_$v.json 2> gpurun_out/exp_search/v_$v.err || { echo "variant $v failed"; tail -3 gpurun_out/exp_search/v_$v.err; continue; }
  python - gpurun_out/exp_search/v_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
l = d["launches_of_one_alignment"]
print("variant", sys.argv[2], "us", [x["us"] for x in l[:8]], "searched", [x["searched_points"] for x in l[:4]])
PY
done
cp /tmp/lib_keep.so sycl_points_amd/lib/libsycl_points_amd.so
