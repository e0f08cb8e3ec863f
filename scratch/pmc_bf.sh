#!/bin/bash
# usage (GPU box, repo root): bash scratch/pmc_bf.sh  -> counters of knn_bf_chunkmin_mfma_kernel (100k x 100k, k = 1)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1)); rm -rf /tmp/pb_$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pb_$i -- python3 $GRAFT_REPO_ROOT/scratch/bf_time.py 1 > /tmp/pb_$i.log 2>&1 || { echo "pass $i failed"; tail -3 /tmp/pb_$i.log; }
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pb_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "chunkmin_mfma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print("%-32s n=%d mean=%.5g" % (c, len(v), sum(v) / len(v)))
PY
