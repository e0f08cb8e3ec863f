#!/bin/bash
# kernel timeline of the last GridKNN.build of scratch/grid_only.py (rocprofv3 --kernel-trace); GPU box, repo root
cd $GRAFT_REPO_ROOT
rm -rf /tmp/tr_grid
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_grid -- python3 $GRAFT_REPO_ROOT/scratch/grid_only.py > /dev/null 2>&1)
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/tr_grid/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
last=max(i for i,n in enumerate(names) if "bbox_kernel" in n)
t0=int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    print("%8.1f us  +%7.1f  %s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Kernel_Name"][:90]))
PY
