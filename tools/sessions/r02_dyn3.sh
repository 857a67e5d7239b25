#!/bin/bash
set -u
O=gpurun_out/r02; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
MI_PT_DYN=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "per_path or render_equals or corpus or feature or numeric or wide_node or phong or large_procedural" > $O/dyn_tests.log 2>&1; rc=$?; echo "pytest (MI_PT_DYN=1) rc=$rc"; tail -3 $O/dyn_tests.log
[ $rc = 0 ] || exit 1
b() { local s=$1 w=$2 h=$3 spp=$4; shift 4; env "$@" timeout -k 10 300 python bench.py --scene $s --width $w --height $h --spp $spp --max-path 0 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['roofline']['terms']; print('%-20s %-12s %8.1f Msamples/s  lds %5d B  eff closest %.2f shadow %.2f' % ('$s', '$*', d['value'], d['config']['launch']['lds_bytes_per_workgroup'], t['simd_efficiency_closest_traversal'] or 0, t['simd_efficiency_shadow_traversal'] or 0))"; }
for d in 0 1; do b LivingRoomLit 1920 1080 64 MI_PT_DYN=$d; done
for d in 0 1; do b MetalRings 1920 1080 64 MI_PT_DYN=$d; done
for d in 0 1; do b CornellBoxSpecular 1024 1024 128 MI_PT_DYN=$d; done
for d in 0 1; do b atrium 1920 1080 32 MI_PT_DYN=$d; done
for d in 0 1; do b clutter 1920 1080 32 MI_PT_DYN=$d; done
