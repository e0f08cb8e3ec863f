#!/bin/bash
# SQ / TA / L2 counters of gicp_align_kernel per launch class, on the GPU box from the repo root:
#   bash profiles/collect_sq.sh <tag>     -> gpurun_out/profiles_<tag>/<tag>_sq_counters_by_launch_class.txt
# Four separate --pmc passes over the same command (counters never share a run with --stats or a trace domain other than
# the kernel trace). Launch k of an alignment = the k-th dispatch modulo 20 of the first five alignments the command runs
# (warm-up, three timed blocks, the correctness alignment: all 20 launches long, criteria 0).
set -e
TAG=${1:-rXX}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/sq_$TAG_p$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sq_${TAG}_p$i -- python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 20 --repeats 3 "${@:2}" > $OUT/sq_p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$TAG" "$OUT" <<'PY'
import collections, csv, glob, sys
tag, out = sys.argv[1], sys.argv[2]
cls = lambda k: "launch 0" if k == 0 else "launch 1" if k == 1 else "launch 2" if k == 2 else "launch 3" if k == 3 else "steady (4-19)"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/sq_{tag}_p*/**/*counter_collection.csv", recursive=True):
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "gicp_align_kernel" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    for i, d in enumerate(sorted(rows)[:100]):
        for c, v in rows[d].items():
            acc[cls(i % 20)][c].append(v)
with open(f"{out}/{tag}_sq_counters_by_launch_class.txt", "w") as o:
    o.write("gicp_align_kernel, rocprofv3 --pmc (4 passes), mean per launch over the first five 20-launch alignments of bench.py\n")
    names = sorted({c for d in acc.values() for c in d})
    o.write("%-26s" % "counter" + "".join("%16s" % k for k in sorted(acc)) + "\n")
    for c in names:
        o.write("%-26s" % c + "".join("%16.4g" % (sum(acc[k][c]) / max(1, len(acc[k][c]))) for k in sorted(acc)) + "\n")
print(open(f"{out}/{tag}_sq_counters_by_launch_class.txt").read())
PY
