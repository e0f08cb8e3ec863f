#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for qt in 2 4 8; do for cpw in 1 2 4 7 14; do
  rm -rf /tmp/bfprof
  SP_EXP_QT=$qt SP_EXP_CPW=$cpw rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bfprof -- python3 scratch/bf_time.py 1 > /tmp/bf.log 2>&1
  f=$(find /tmp/bfprof -name "*kernel_stats.csv" | head -1)
  echo "qt=$qt cpw=$cpw $(grep -v '^W\|^E' /tmp/bf.log | tail -1) mfma_us=$(grep chunkmin_mfma $f | awk -F, '{print $(NF-5)}' | head -1)"
done; done
