#!/bin/bash
# A/B on ONE box: bench.py with every lib under scratch/ab/*.so, interleaved twice; prints the figures that matter.
# usage (on the GPU box, repo root): bash scratch/ab.sh [extra bench args]
mkdir -p gpurun_out
for rep in 1 2; do
for lib in scratch/ab/*.so; do
  SP_AMD_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-stages "$@" > gpurun_out/ab_$(basename $lib .so).r$rep.log 2>&1
  python - "$lib" "gpurun_out/ab_$(basename $lib .so).r$rep.log" <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[2]) if x.startswith("{")][-1])
    L = [x["us"] for x in d["launches_of_one_alignment"]]
    print(f"{sys.argv[1]:40s} step {1e3*d['ms_per_step']:.2f} us  launches {L[0]:.1f} {L[1]:.1f} {L[2]:.1f} {L[3]:.1f} steady {sorted(L[4:])[len(L[4:])//2]:.2f}  "
          f"until-converged {1e3*d['until_converged']['ms_per_alignment']:.1f} us  prepare {1e3*d['kernels']['source_prepare']['ms']:.1f}  pose err {d['pose_max_abs_err_vs_ground_truth']:.2e}", flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]).read()[-500:])
PY
done; done
