#!/bin/bash
# same box: round-2 bench (library + python of commit 08b158f) against the working tree, alternating
cd $GRAFT_REPO_ROOT
for tag in old new old new; do
  if [ $tag = old ]; then f=scratch/old_r02/repo/bench.py; else f=bench.py; fi
  extra=""; [ $tag = new ] && extra="--no-stages"
  python $f --no-cpu-baseline $extra > gpurun_out/abb_$tag.log 2>/dev/null
  python - <<PY
import json
for l in open("gpurun_out/abb_$tag.log"):
    if l.startswith("{"):
        d=json.loads(l); lc=d.get("launch_classes") or {}
        print("$tag", round(d["ms_per_step"]*1e3,2), "us/step; kernel", round(d["kernels"]["gicp_align_kernel"]["ms"]*1e3,2), "; steady", (lc.get("steady") or {}).get("mean_us"), "; searching", (lc.get("searching") or {}).get("mean_us"), "; until_converged", (d.get("until_converged") or {}).get("ms_per_alignment"))
PY
done
