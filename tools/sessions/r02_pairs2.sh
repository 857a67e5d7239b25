#!/bin/bash
# pair leaves everywhere (LDS copy + quantised records): full GPU suite, then A/B on scenes read from HBM
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pairs_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pairs_tests.log
[ $rc = 0 ] || exit 1
b() { local s=$1 w=$2 h=$3 spp=$4 mp=$5; shift 5; env "$@" timeout -k 10 300 python bench.py --scene $s --width $w --height $h --spp $spp --max-path $mp --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['roofline']['terms']; print('%-20s %-14s %8.1f Msamples/s  N %.2f T %.2f N\' %.2f T\' %.2f' % ('$s', '$*', d['value'], t['N'], t['T'], t['N_shadow_per_segment'], t['T_shadow_per_segment']))"; }
for e in MI_PT_PAIRS=0 MI_PT_PAIRS=1; do
  b CornellBoxDiffuse 512 512 1024 8 $e
  b TestCase8 512 512 256 0 $e
  b CornellBoxSpecular 1024 1024 128 0 $e
  b LivingRoomLit 1920 1080 64 0 $e
  b MetalRings 1920 1080 64 0 $e
  b atrium 1920 1080 64 0 $e
  b clutter 1920 1080 32 0 $e
done 2>&1 | tee $O/pairs_ab.txt
