#!/bin/bash
# experiment builds of registration.hip (SP_DEV_MIN: one instantiation, ~30 s each), linked against the shipped objects:
#   bash scratch/exp_build.sh "<flags of variant 0>" "<flags of variant 1>" ...   -> scratch/exp_libs/lib_<i>.so
cd /root/repo/sycl_points_amd/csrc
mkdir -p /tmp/exp /root/repo/scratch/exp_libs
i=0
for flags in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -DSP_DEV_MIN $flags \
      -Rpass-analysis=kernel-resource-usage -c registration.hip -o /tmp/exp/registration_$i.o 2> /tmp/exp/remarks_$i.txt \
    && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /root/repo/scratch/exp_libs/lib_$i.so $(ls build/*.o | grep -vFx build/registration.o) /tmp/exp/registration_$i.o -ldl \
    || { grep -E "error" -A6 /tmp/exp/remarks_$i.txt | head -30; } ) &
  i=$((i+1))
done
wait
for j in $(seq 0 $((i-1))); do
  echo "variant $j:"; grep -E "Function Name|VGPRs:|ScratchSize|VGPRs Spill" /tmp/exp/remarks_$j.txt | sed 's/.*remark: *//; s/ *\[-Rpass.*//' | paste - - - - | grep -E "align_kernel|persistent" | sed 's/_ZN2sp12_GLOBAL__N_1//'
done
