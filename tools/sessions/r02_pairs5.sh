#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O
b() { local s=$1 w=$2 h=$3 spp=$4 mp=$5; shift 5; env "$@" timeout -k 10 300 python bench.py --scene $s --width $w --height $h --spp $spp --max-path $mp --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['roofline']['terms']; print('%-20s %-60s %8.1f Msamples/s' % ('$s', '$*', d['value']))"; }
for e in "MI_PT_PAIRS=0 MI_PT_LIB=$PWD/master_amd/libmi_pt_nopw.so" "MI_PT_PAIRS=0 A=1" "MI_PT_PAIRS=1 A=1"; do
  b atrium 1920 1080 256 0 $e
  b LivingRoomLit 1920 1080 128 0 $e
done 2>&1 | tee $O/pairs_ab3.txt
