#!/bin/bash
# per-rank step time of the emulated W-rank run for two dealings of the source, every rank (GPU box, repo root)
cd $GRAFT_REPO_ROOT
for W in 8 4 2; do
for c in 1024 500000; do
  for r in $(seq 0 $((W-1))); do
    python bench.py --emulate-world $W --emulate-rank $r --shard-chunk $c --no-cpu-baseline --no-stages --repeats 7 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); l=d['launches_of_one_alignment']
print('W', $W, 'chunk', $c, 'rank', $r, 'us/step', round(d['ms_per_step']*1e3,2), [round(x['us']) for x in l[:8]])"
  done
done
done
