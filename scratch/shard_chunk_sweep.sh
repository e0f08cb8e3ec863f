#!/bin/bash
# per-rank step time of the emulated 8-rank run by chunk size of the source dealing and by rank (GPU box, repo root)
cd $GRAFT_REPO_ROOT
W=${1:-8}
for c in 1024 16384 62500 125000 250000 500000 0; do
  for r in 0 3 7; do
    python bench.py --emulate-world $W --emulate-rank $r --shard-chunk $c --no-cpu-baseline --no-stages --repeats 7 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); l=d['launches_of_one_alignment']
print('chunk', $c, 'rank', $r, 'us/step', round(d['ms_per_step']*1e3,2), [round(x['us']) for x in l[:8]], 'pose err %.2e' % d['pose_max_abs_err_vs_ground_truth'])"
  done
done
