#!/bin/bash
# usage (GPU box, repo root): bash scratch/pmc_bvh.sh <k> [old]  -> counters of the self-kNN kernel on 1M uniform points (heap kernel; "old": sorted insertion)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1)); rm -rf /tmp/pb_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pb_$i -- python3 $GRAFT_REPO_ROOT/scratch/bvh_profile.py $1 $2 > /dev/null 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pb_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bvh_search_kernel" in r["Kernel_Name"] or "bvh_heap_kernel" in r["Kernel_Name"]:
            acc[("heap " if "heap" in r["Kernel_Name"] else "sorted-insertion ") + r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print("%-44s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
