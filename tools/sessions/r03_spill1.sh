#!/bin/bash
# r03: register budget of the dynamic-fetch kernels (HBM-resident scenes): 6 / 5 / 4 waves per SIMD = 80 / 96 / 128 VGPRs
set -o pipefail
mkdir -p gpurun_out
run() { MI_PT_LIB=$PWD/master_amd/$6 timeout -k 10 300 python bench.py --scene "$1" --width $2 --height $3 --spp $4 --max-path $5 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-20s %-18s %7.0f Msamples/s %8.1f ms' % ('$6', '$1', d['value'], d['ms_per_step']))"; }
for lib in ${LIBS:-libmi_pt.so}; do
  run atrium 1920 1080 128 0 $lib
  run atrium:2000000 1920 1080 64 0 $lib
  run clutter 1920 1080 64 0 $lib
  run LivingRoomLit 1920 1080 128 0 $lib
  run MetalRings 1920 1080 128 0 $lib
  run CornellBoxSpecular 1024 1024 256 0 $lib
done 2>&1 | tee gpurun_out/r03_spill1.txt
