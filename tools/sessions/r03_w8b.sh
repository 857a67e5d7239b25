#!/bin/bash
mkdir -p gpurun_out
for W8 in 0 1; do
  MI_PT_WIDE8=$W8 python bench.py --scene atrium --width 960 --height 540 --spp 32 --max-path 0 --steps 1 --warmup 1 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['roofline']['terms']; print('W8=$W8', round(d['value']), {k:(round(v,2) if isinstance(v,float) else v) for k,v in t.items()})"
done
