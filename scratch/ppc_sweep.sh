#!/bin/bash
# usage (GPU box, repo root): bash scratch/ppc_sweep.sh <tag> -> gpurun_out/<tag>_ppc_*.json : the bench line at several grid densities
for ppc in 0.5 1.0 2.0 3.0; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --ppc $ppc --repeats 5 > gpurun_out/$1_ppc_$ppc.json 2> gpurun_out/$1_ppc_$ppc.err || echo "ppc $ppc failed"
done
python - "$1" <<'PY'
import json, sys
for ppc in ("0.5", "1.0", "2.0", "3.0"):
    try:
        d = json.load(open(f"gpurun_out/{sys.argv[1]}_ppc_{ppc}.json"))
    except Exception as e:
        print(ppc, "no line", e); continue
    l = d["launches_of_one_alignment"]
    print(ppc, "ms/step %.4f" % d["ms_per_step"], "until_conv %.3f" % d["until_converged"]["ms_per_alignment"],
          " | ".join("%d: %.0f+%.0f (%d)" % (x["k"], x["search_launch_us"], x["streaming_launch_us"], x["searched_points"]) for x in l[:5]))
PY
