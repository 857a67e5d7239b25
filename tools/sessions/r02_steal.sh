#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_cadence_c5.py tests/test_gpu_parity.py -m gpu -x -q -k "dynamic_fetch or per_path or render_equals or corpus or wide_node or phong or large_procedural or intersect" > gpurun_out/r02/steal_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r02/steal_tests.log
[ $rc = 0 ] || exit 1
b() { local s=$1 w=$2 h=$3 spp=$4; shift 4; env "$@" timeout -k 10 300 python bench.py --scene $s --width $w --height $h --spp $spp --max-path 0 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['roofline']['terms']; print('%-20s %-46s %8.1f Msamples/s  unified eff %.3f' % ('$s', '$*'.replace('$PWD/master_amd/',''), d['value'], t.get('simd_efficiency_unified_traversal') or 0))"; }
for cfg in "LivingRoomLit 1920 1080 64" "MetalRings 1920 1080 64" "atrium 1920 1080 32" "clutter 1920 1080 32" "CornellBoxSpecular 1024 1024 128" "atrium:2000000 1920 1080 16"; do
  b $cfg A=steal
  b $cfg MI_PT_LIB=$PWD/master_amd/libmi_pt_nosteal.so
done
