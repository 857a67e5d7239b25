#!/bin/bash
set -u
O=gpurun_out/r02; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
b() { env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-hbm-workload 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-40s %.1f Msamples/s  lds %d B' % ('$*', d['value'], d['config']['launch']['lds_bytes_per_workgroup']))"; }
b MI_PT_DYN=0
b MI_PT_DYN=0 MI_PT_LDS_PAD=7232
b MI_PT_DYN=0 MI_PT_LDS_PAD=16000
b MI_PT_DYN=1
echo "--- phase stamps, classic"; MI_PT_DYN=0 python tools/gpu_phase.py 2>&1 | head -10
echo "--- phase stamps, dynamic fetch"; MI_PT_DYN=1 python tools/gpu_phase.py 2>&1 | head -10
