#!/bin/bash
# A/B of a scratch library against the shipped one on one box: headline launches + the hard_init cases of bench.py.
# usage (GPU box, repo root): bash scratch/ab_hard.sh scratch/ab/lib_x.so
cd $GRAFT_REPO_ROOT
LIB=$1
cp sycl_points_amd/lib/libsycl_points_amd.so /tmp/lib_keep.so
for tag in A B A B; do
  if [ $tag = A ]; then cp /tmp/lib_keep.so sycl_points_amd/lib/libsycl_points_amd.so; else cp $LIB sycl_points_amd/lib/libsycl_points_amd.so; fi
  timeout -k 10 200 python bench.py --repeats 7 --no-cpu-baseline > /tmp/ab_$tag.json 2> /tmp/ab_$tag.err || { echo "$tag failed"; tail -3 /tmp/ab_$tag.err; continue; }
  python - /tmp/ab_$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
l = d["launches_of_one_alignment"]
print(sys.argv[2], "launch us", [x["us"] for x in l[:5]], "us/step", round(d["ms_per_step"] * 1e3, 2), "conv ms", round(d["until_converged"]["ms_per_alignment"], 4),
      "hard", {k: round(v["ms_per_alignment"], 3) for k, v in d["hard_init"].items()}, "p2d", round(d["point_to_distribution"]["ms_per_step"] * 1e3, 2))
PY
done
cp /tmp/lib_keep.so sycl_points_amd/lib/libsycl_points_amd.so
