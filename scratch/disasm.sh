#!/bin/bash
# disassemble one kernel of build/registration.o: scratch/disasm.sh <mangled-name-substring> > out.s
cd /root/repo/sycl_points_amd/csrc/build
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading registration.o >/dev/null 2>&1
/opt/rocm/lib/llvm/bin/llvm-objdump -d registration.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 2>/dev/null | awk -v pat="$1" '/^[0-9a-f]+ <.*>:/{p = index($0, pat) > 0} p'
