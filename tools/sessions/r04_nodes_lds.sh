#!/bin/bash
# wide records in LDS (MI_PT_NODES_LDS_KB: 0 = read from HBM; limit of the record set staged) on the models whose records fit
run() { env MI_PT_NODES_LDS_KB=$1 python bench.py --scene $2 --width $3 --height $4 --spp $5 --max-path 0 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse --no-fast-variant --no-live-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NODES_LDS_KB=$1 %-20s %8.1f Msamples/s %9.2f ms  lds %s  %s' % ('$2', d['value'], d['ms_per_step'], d['config']['launch']['lds_bytes_per_workgroup'], d['roofline']['kernel'][-44:]))"; }
for rep in 1 2; do for kb in 0 64; do
run $kb CornellBoxSpecular 1024 1024 256
run $kb MirrorBalls 1024 1024 256
run $kb SimpleSphereIOR2 1024 1024 256
done; done
