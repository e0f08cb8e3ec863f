#!/bin/bash
# development build: one instantiation of the per-iteration kernel (registration.hip in ~30 s instead of ~4 min)
set -e
cd /root/repo/sycl_points_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -DSP_DEV_MIN "$@" -Rpass-analysis=kernel-resource-usage -c registration.hip -o build/registration.o 2> /tmp/reg_dev.remarks || { grep -E "error" -A8 /tmp/reg_dev.remarks | head -40; exit 1; }
grep -E "Function Name|VGPRs:|ScratchSize|VGPRs Spill|LDS Size|SGPRs:" /tmp/reg_dev.remarks | sed 's/.*remark: *//; s/ *\[-Rpass.*//' | paste - - - - - - | grep align_kernel
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libsycl_points_amd.so build/*.o -ldl
echo linked
