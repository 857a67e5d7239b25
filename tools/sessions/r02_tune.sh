#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O
b() { local s=$1 w=$2 h=$3 spp=$4 mp=$5; shift 5; env "$@" timeout -k 10 300 python bench.py --scene $s --width $w --height $h --spp $spp --max-path $mp --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-workload 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-20s %-22s %8.1f Msamples/s' % ('$s', '$*', d['value']))"; }
for c in 26 32 41 52 64; do b CornellBoxDiffuse 512 512 1024 8 MI_PT_CHUNK_SPP=$c; done 2>&1 | tee $O/tune_chunk.txt
timeout -k 10 300 python tests/tools/lds_limit.py 2>&1 | tee $O/tune_lds_limit.txt
