#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
b() { local s=$1 w=$2 h=$3 spp=$4; shift 4; env "$@" timeout -k 10 300 python bench.py --scene $s --width $w --height $h --spp $spp --max-path 0 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['roofline']['terms']; print('%-20s %-40s %8.1f Msamples/s  lds %5d  N %.1f eff %.3f' % ('$s', '$*', d['value'], d['config']['launch']['lds_bytes_per_workgroup'], t['N'], t.get('simd_efficiency_unified_traversal') or 0))"; }
for cfg in "LivingRoomLit 1920 1080 64" "MetalRings 1920 1080 64" "CornellBoxSpecular 1024 1024 128" "TestCase8 512 512 256" "Door 1024 1024 64" "TestStand 1024 1024 64"; do
  b $cfg A=default
  b $cfg MI_PT_WIDE_NODES=1
  b $cfg MI_PT_WIDE_NODES=1 MI_PT_FLOAT_NODES=0
done
