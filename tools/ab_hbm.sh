#!/bin/bash
# A/B of library builds on the scenes read from HBM: one line per (lib, scene)
LIBS="$@"; for rep in 1 2; do for lib in $LIBS; do for spec in "atrium 1920 1080 128" "clutter 1920 1080 64" "atrium:2000000 1920 1080 32" "CornellBoxSpecular 1024 1024 256" "LivingRoomLit 1920 1080 64"; do set -- $spec; 
MI_PT_LIB=$PWD/master_amd/$lib python bench.py --scene $1 --width $2 --height $3 --spp $4 --max-path 0 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse --no-fast-variant --no-live-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib %-22s %8.1f Msamples/s %9.2f ms' % ('$1', d['value'], d['ms_per_step']))"; done; done; done
