#!/bin/bash
# usage: scratch/sweep.sh <outdir> <flag-name> v1 v2 ...   -> per-launch profile + until-converged for bench.py --<flag-name> v
OUT=gpurun_out/$1; FLAG=$2; shift 2
mkdir -p $OUT
for v in "$@"; do
  timeout -k 10 150 python bench.py --$FLAG $v --repeats 3 --no-cpu-baseline $SWEEP_EXTRA > $OUT/${FLAG}_$v.json 2> $OUT/${FLAG}_$v.err || { echo "$FLAG $v failed"; tail -3 $OUT/${FLAG}_$v.err; continue; }
  python - $OUT/${FLAG}_$v.json "$FLAG $v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], [l["us"] for l in d["launches_of_one_alignment"][:5]], "steady", d["launch_classes"]["steady"]["mean_us"],
      "conv ms", round(d["until_converged"]["ms_per_alignment"], 4), "it", d["until_converged"]["iterations_executed"],
      "ms/step", round(d["ms_per_step"], 5))
PY
done
