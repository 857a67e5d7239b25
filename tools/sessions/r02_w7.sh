#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
b() { local s=$1 w=$2 h=$3 spp=$4; shift 4; env "$@" timeout -k 10 300 python bench.py --scene $s --width $w --height $h --spp $spp --max-path 0 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-20s %-66s %8.1f Msamples/s  lds %5d' % ('$s', '$*'.replace('$PWD/master_amd/',''), d['value'], d['config']['launch']['lds_bytes_per_workgroup']))"; }
for cfg in "LivingRoomLit 1920 1080 64" "atrium 1920 1080 32" "clutter 1920 1080 32" "CornellBoxSpecular 1024 1024 128"; do
  b $cfg A=default
  b $cfg MI_PT_STACK_ROWS=8
  b $cfg MI_PT_STACK_ROWS=8 MI_PT_LIB=$PWD/master_amd/libmi_pt_w7.so
  b $cfg MI_PT_STACK_ROWS=4 MI_PT_LIB=$PWD/master_amd/libmi_pt_w7.so
done
