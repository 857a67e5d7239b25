#!/bin/bash
# MetalRings (coarse 16-bit grid: far light quads stretch the scene box): full-precision binary nodes (default) against the wide quantised records, r04 kernels
run() { env "$@" python bench.py --scene MetalRings --width 1920 --height 1080 --spp 64 --max-path 0 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse --no-fast-variant --no-live-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['roofline']['terms']; print('$*  %8.1f Msamples/s  N %.2f N_shadow %.2f T %.2f  %s' % (d['value'], t['N'], t['N_shadow_per_segment'], t['T'], d['roofline']['kernel'][-50:]))"; }
for rep in 1 2; do
run X=0
run MI_PT_WIDE_NODES=1
run MI_PT_WIDE_NODES=1 MI_PT_DYN_UNI=0
run MI_PT_WIDE_NODES=0 MI_PT_FLOAT_NODES=0
done
