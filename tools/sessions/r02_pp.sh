#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bpt.py -m gpu -x -q > gpurun_out/r02/pp_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 gpurun_out/r02/pp_tests.log
[ $rc = 0 ] || exit 1
b() { local s=$1 w=$2 h=$3 spp=$4 mp=$5; shift 5; env "$@" timeout -k 10 300 python bench.py --scene $s --width $w --height $h --spp $spp --max-path $mp --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-workload 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-20s %-46s %8.1f Msamples/s' % ('$s', '$*'.replace('$PWD/master_amd/',''), d['value']))"; }
for i in 1 2; do
b CornellBoxDiffuse 512 512 1024 8 A=postpone
b CornellBoxDiffuse 512 512 1024 8 MI_PT_LIB=$PWD/master_amd/libmi_pt_nopp.so
done
b TestCaseFurnace 512 512 256 0 A=postpone
b TestCaseFurnace 512 512 256 0 MI_PT_LIB=$PWD/master_amd/libmi_pt_nopp.so
b TestCase8 512 512 256 0 A=postpone
b TestCase8 512 512 256 0 MI_PT_LIB=$PWD/master_amd/libmi_pt_nopp.so
b LivingRoomLit 1920 1080 32 0 MI_PT_DYN=0
b LivingRoomLit 1920 1080 32 0 MI_PT_DYN=0 MI_PT_LIB=$PWD/master_amd/libmi_pt_nopp.so
python tools/census.py
