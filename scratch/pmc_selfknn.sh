#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_selfknn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/scratch/selfknn_only.py > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "grid_self_knn_wave_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c,v in sorted(acc.items()):
    print("   %-24s n=%d mean=%.4g"%(c,len(v),sum(v)/len(v)))
PY
