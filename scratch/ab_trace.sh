#!/bin/bash
# same box: per-launch kernel durations (rocprofv3 --kernel-trace) of round 2's library and of the working tree
cd /tmp && export TMPDIR=/tmp
for tag in old new old new; do
  if [ $tag = old ]; then f=$GRAFT_REPO_ROOT/scratch/old_r02/repo/bench.py; extra=""; else f=$GRAFT_REPO_ROOT/bench.py; extra="--no-stages"; fi
  rm -rf /tmp/abt
  rocprofv3 --kernel-trace --output-format csv -d /tmp/abt -- python3 $f --no-cpu-baseline $extra --steps 20 --warmup 20 > /tmp/abt.log 2>&1
  python3 - $tag <<'PY'
import csv, glob, sys, collections
f = glob.glob("/tmp/abt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
al = [r for r in rows if "gicp_align_kernel" in r["Kernel_Name"]]
al.sort(key=lambda r: int(r["Start_Timestamp"]))
al = al[20:20 + 31 * 20]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in al]
n = len(d) // 20
per = [sum(d[a * 20 + k] for a in range(n)) / n for k in range(20)]
gaps = [(int(al[i + 1]["Start_Timestamp"]) - int(al[i]["End_Timestamp"])) / 1e3 for i in range(len(al) - 1) if (i % 20) != 19]
other = collections.Counter()
for r in rows:
    if "gicp_align_kernel" not in r["Kernel_Name"]:
        other[r["Kernel_Name"][:60]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(sys.argv[1], "launch0-2: %.1f %.1f %.1f  steady mean %.2f  all mean %.2f  gap between launches %.2f" % (per[0], per[1], per[2], sum(per[3:]) / 17, sum(d) / len(d), sum(gaps) / len(gaps)))
PY
done
