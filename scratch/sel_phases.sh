#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for stop in 1 2 3 4 5 0; do
  rm -rf /tmp/skp
  SP_TMP_STOP=$stop rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/skp -- python3 $GRAFT_REPO_ROOT/scratch/selfknn_one.py 20 0 > /tmp/sk.log 2>&1
  f=$(find /tmp/skp -name "*kernel_stats.csv" | head -1)
  echo "stop=$stop select_us=$(grep select_kernel $f | awk -F, '{print $(NF-4)}')"
done
