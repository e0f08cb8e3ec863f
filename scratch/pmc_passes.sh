#!/bin/bash
# usage: scratch/pmc_passes.sh <outdir-name>   (run on the GPU box from the repo root)
set -e
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --no-cpu-baseline --steps 40 --warmup 20 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        for name in ("gicp_align_kernel","align_finish","gicp_fused_kernel"):
            if name in k: acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn,d in acc.items():
    print(kn)
    for c,v in sorted(d.items()):
        v2=sorted(v); 
        print("   %-24s n=%d mean=%.4g median=%.4g min=%.4g max=%.4g"%(c,len(v),sum(v)/len(v),v2[len(v2)//2],v2[0],v2[-1]))
PY
