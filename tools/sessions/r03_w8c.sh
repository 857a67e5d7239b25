#!/bin/bash
mkdir -p gpurun_out
run() { MI_PT_LIB=$PWD/master_amd/$6 timeout -k 10 300 python bench.py --scene "$1" --width $2 --height $3 --spp $4 --max-path $5 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-20s W8=%s %-18s %7.0f Msamples/s %8.1f ms' % ('$6', '$MI_PT_WIDE8', '$1', d['value'], d['ms_per_step']))"; }
for lib in libmi_pt.so libmi_pt_dyn5.so; do for W8 in 0 1; do export MI_PT_WIDE8=$W8
  run atrium 1920 1080 128 0 $lib
  run atrium:2000000 1920 1080 64 0 $lib
  run atrium:8000000 1920 1080 32 0 $lib
  run clutter 1920 1080 64 0 $lib
  run LivingRoomLit 1920 1080 128 0 $lib
done; done 2>&1 | tee gpurun_out/r03_w8c.txt
