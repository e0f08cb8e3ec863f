#!/bin/bash
# timeline of ONE loop of the reference's example through the facade (the last of 6): every kernel / copy with its start
# relative to the loop's first kernel, its duration and the gap before it; GPU box
cd $GRAFT_REPO_ROOT/tests/cpp
rm -rf /tmp/prof_tl
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_tl -- $GRAFT_REPO_ROOT/tests/cpp/example_registration $GRAFT_REPO_ROOT/tests/golden/source.ply $GRAFT_REPO_ROOT/tests/golden/target.ply 6 0 > /dev/null 2>&1)
python3 - <<PY
import csv, glob
ev = []
for f in glob.glob("/tmp/prof_tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for f in glob.glob("/tmp/prof_tl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", ""))[:40] + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# loops start with the big host-to-device copy of the source cloud
starts = [i for i, e in enumerate(ev) if e[2].startswith("COPY") and "HOST_TO_DEVICE" in e[2] and e[2].split()[-1].isdigit() and int(e[2].split()[-1]) > 1000000]
i0 = starts[-2] if len(starts) >= 2 else 0
if not starts:  # the box filter reads the scan in place (no upload): a loop starts at the launch before its first compaction
    comp = [i for i, e in enumerate(ev) if "compact_fused" in e[2]]
    i0 = max(comp[-4] - 1, 0) if len(comp) >= 4 else 0
t0 = ev[i0][0]
prev = t0
for s, e, n in ev[i0:]:
    print("%9.1f us  dur %7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n))
    prev = e
PY
