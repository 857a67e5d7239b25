#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { MI_PT_Q4_ORDER=$6 timeout -k 10 300 python bench.py --scene "$1" --width $2 --height $3 --spp $4 --max-path $5 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order=%s %-18s %7.0f Msamples/s %8.1f ms' % ('$6', '$1', d['value'], d['ms_per_step']))"; }
for o in 0 1 2; do
  run atrium 1920 1080 128 0 $o
  run atrium:2000000 1920 1080 64 0 $o
  run atrium:8000000 1920 1080 32 0 $o
  run clutter 1920 1080 64 0 $o
  run LivingRoomLit 1920 1080 64 0 $o
done 2>&1 | tee gpurun_out/r03_q4order.txt
