#!/bin/bash
# A/B of an environment switch on the scenes read from HBM: tools/ab_env.sh VAR valueA valueB   (one line per value and scene, two repeats)
VAR=$1; shift
VALS="$@"; for rep in 1 2; do for val in $VALS; do for spec in "atrium 1920 1080 128" "clutter 1920 1080 64" "atrium:2000000 1920 1080 32" "CornellBoxSpecular 1024 1024 256" "LivingRoomLit 1920 1080 64" "MetalRings 1920 1080 64"; do set -- $spec
env $VAR=$val python bench.py --scene $1 --width $2 --height $3 --spp $4 --max-path 0 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse --no-fast-variant --no-live-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['roofline']['terms']; print('$VAR=$val %-20s %8.1f Msamples/s %9.2f ms   N %.2f N_shadow %.2f T %.2f' % ('$1', d['value'], d['ms_per_step'], t['N'], t['N_shadow_per_segment'], t['T']))"; done; done; done
