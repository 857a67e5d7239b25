#!/bin/bash
run() { timeout -k 10 200 python bench.py --scene "$1" --width $2 --height $3 --spp $4 --max-path $5 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],1))"; }
run CornellBoxDiffuse 512 512 1024 8
run CornellBoxSpecular 1024 1024 512 0
run LivingRoomLit 1920 1080 128 0
run MetalRings 1920 1080 128 0
run atrium 1920 1080 128 0
run clutter 3840 2160 64 0
run atrium:2000000 1920 1080 64 0
python tools/bpt_prof.py CornellBoxDiffuse 2>&1 | tail -1
python tools/bpt_prof.py LivingRoomLit 2>&1 | tail -1
