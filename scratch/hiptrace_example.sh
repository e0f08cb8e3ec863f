#!/bin/bash
# host-side HIP API timeline of ONE loop of the reference's example (the last of 6): which runtime calls the loop's host time goes to
cd $GRAFT_REPO_ROOT/tests/cpp
rm -rf /tmp/prof_api
(cd /tmp && TMPDIR=/tmp rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/prof_api -- $GRAFT_REPO_ROOT/tests/cpp/example_registration $GRAFT_REPO_ROOT/tests/golden/source.ply $GRAFT_REPO_ROOT/tests/golden/target.ply 6 0 > /dev/null 2>&1)
python3 - <<PY
import csv, glob, collections
api = []
for f in glob.glob("/tmp/prof_api/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
api.sort()
ker = []
for f in glob.glob("/tmp/prof_api/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ker.append((int(r["Start_Timestamp"]), r["Kernel_Name"][:50]))
ker.sort()
# the last loop: from the last-but-one big H2D memcpy call
big = [i for i, a in enumerate(api) if a[2].startswith("hipMemcpy") and a[1] - a[0] > 15000]
opt = [k for k in ker if "gicp_optimize_kernel" in k[1]]
t_end = api[-1][1]
t0 = [k for k in ker if "compact_fused" in k[1]][-2][0] - 150000
print("API calls of the last loop (start us | dur us | gap before us | name)")
prev = None
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in api:
    if s < t0: continue
    gap = 0 if prev is None else (s - prev) / 1e3
    print("%9.1f %7.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
    tot[n] += (e - s) / 1e3; cnt[n] += 1
    prev = e
print("--- totals")
for n, v in tot.most_common():
    print("%8.1f us  %4d  %s" % (v, cnt[n], n))
PY
