#!/bin/bash
# kernels / copies per loop of the reference's example (tests/cpp/example_registration, 10 loops): rocprofv3 --stats; GPU box
cd $GRAFT_REPO_ROOT/tests/cpp
rm -rf /tmp/prof_ex
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_ex -- $GRAFT_REPO_ROOT/tests/cpp/example_registration $GRAFT_REPO_ROOT/tests/golden/source.ply $GRAFT_REPO_ROOT/tests/golden/target.ply 10 0 $EXTRA > /dev/null 2>&1)
python3 - <<PY
import csv, glob
for pat in ("*kernel_stats.csv", "*memory_copy_stats.csv"):
    for f in glob.glob("/tmp/prof_ex/**/" + pat, recursive=True):
        print(pat)
        for r in list(csv.DictReader(open(f)))[:22]:
            print("  %-70s calls/loop %6.1f  avg us %7.1f" % (r["Name"][:70], int(r["Calls"]) / 10.0, float(r["AverageNs"]) / 1e3))
PY
