#!/bin/bash
# launch-0 time under the variants of the search kernel (debug bits of split_first)
for v in "$@"; do
  python bench.py --no-cpu-baseline --no-stages --internal split_first=$v > gpurun_out/st_$v.json 2> gpurun_out/st_$v.err
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads([x for x in open(f"gpurun_out/st_{v}.json") if x.startswith("{")][-1])
    L = [x["us"] for x in d["launches_of_one_alignment"]]
    print(f"split_first={v}: launch 0 {L[0]:.1f} us, launch 1 {L[1]:.1f}, step {1e3*d['ms_per_step']:.2f}", flush=True)
except Exception as e:
    print(v, "FAILED", e, open(f"gpurun_out/st_{v}.err").read()[-500:])
PY
done
