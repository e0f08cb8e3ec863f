#!/bin/bash
# per-launch durations of gicp_align_kernel over one alignment, for scratch/lib_A.so and scratch/lib_B.so
cd $GRAFT_REPO_ROOT
for tag in A B; do
  cp scratch/lib_$tag.so sycl_points_amd/lib/libsycl_points_amd.so
  rm -rf /tmp/tr_$tag
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 20 --no-cpu-baseline $@ > /dev/null 2>&1)
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/tr_$tag/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "gicp_align_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
print("$tag", [round(x,1) for x in d[20:40]])
PY
done
