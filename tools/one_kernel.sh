#!/bin/bash
# Registers / scratch of ONE megakernel instantiation (no GPU needed):  tools/one_kernel.sh 'false,0,false,6,2,3,true,true,true' [extra flags]
K=$1; shift
cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I /root/repo/include --offload-arch=gfx950 -munsafe-fp-atomics \
  --cuda-device-only -S -o /tmp/one_kernel.s /root/repo/master_amd/csrc/device/pt_kernels.hip "-DMI_ONE_KERNEL=$K" -Rpass-analysis=kernel-resource-usage "$@" 2>&1 \
  | grep -E "VGPRs:|SGPRs:|ScratchSize|Occupancy|error" | sed 's/.*remark: [^ ]* //' | tr '\n' ' '; echo
grep -c "scratch_store" /tmp/one_kernel.s | sed 's/^/scratch_store instructions: /'
