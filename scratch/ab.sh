#!/bin/bash
# A/B on one box: scratch/lib_A.so (before) against scratch/lib_B.so (after); the working library is left at B
cd $GRAFT_REPO_ROOT
for tag in A B A B; do
  cp scratch/lib_$tag.so sycl_points_amd/lib/libsycl_points_amd.so
  python bench.py --no-cpu-baseline --steps 200 --warmup 40 $@ > gpurun_out/ab_$tag.log 2>/dev/null
  python - <<PY
import json
for l in open("gpurun_out/ab_$tag.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$tag", round(d["ms_per_step"]*1e3,2), "us/step  kernel", round(d["kernels"]["gicp_align_kernel"]["ms"]*1e3,2), "pose err", d["pose_max_abs_err_vs_ground_truth"], "inl", d["inliers_last_iteration"])
PY
done
