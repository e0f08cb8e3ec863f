#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gbp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gbp -- python3 $GRAFT_REPO_ROOT/scratch/gridbuild_only.py > /tmp/gb.log 2>&1
tail -3 /tmp/gb.log
f=$(find /tmp/gbp -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"][:80].ljust(82), r["Calls"], round(float(r["AverageNs"])/1e3,2))
PY
