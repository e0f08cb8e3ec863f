#!/bin/bash
cd $GRAFT_REPO_ROOT
rm -rf /tmp/tr_vox
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_vox -- python3 $GRAFT_REPO_ROOT/scratch/voxel_only.py > /dev/null 2>&1)
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/tr_vox/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last call: find the last key_box / key32 kernel and print from there
names=[r["Kernel_Name"] for r in rows]
last=max(i for i,n in enumerate(names) if "voxel_init_kernel" in n)
t0=int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    print("%8.1f us  +%7.1f  %s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Kernel_Name"][:90]))
PY
