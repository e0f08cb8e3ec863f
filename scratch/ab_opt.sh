#!/bin/bash
# A/B of an internal source option on ONE box: bash scratch/ab_opt.sh persistent 0 1
opt=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
for v in "$@"; do
  python bench.py --no-cpu-baseline --no-stages --internal $opt=$v > gpurun_out/abopt_${opt}_$v.r$rep.log 2>&1
  python - "$opt=$v" "gpurun_out/abopt_${opt}_$v.r$rep.log" <<'PY'
import json, sys
try:
    d = json.loads([x for x in open(sys.argv[2]) if x.startswith("{")][-1])
    L = [x["us"] for x in d["launches_of_one_alignment"]]
    print(f"{sys.argv[1]:16s} step {1e3*d['ms_per_step']:.2f} us  per-launch profile {L[0]:.1f} {L[1]:.1f} {L[2]:.1f} {L[3]:.1f} steady {sorted(L[4:])[len(L[4:])//2]:.2f}  "
          f"until-converged {1e3*d['until_converged']['ms_per_alignment']:.1f} us ({d['until_converged']['iterations_executed']} it)  pose err {d['pose_max_abs_err_vs_ground_truth']:.2e}", flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2]).read()[-800:])
PY
done; done
