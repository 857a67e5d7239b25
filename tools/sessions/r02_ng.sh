#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bpt.py -m gpu -x -q > $O/ng_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/ng_tests.log
[ $rc = 0 ] || exit 1
b() { local s=$1 w=$2 h=$3 spp=$4 mp=$5; shift 5; env "$@" timeout -k 10 300 python bench.py --scene $s --width $w --height $h --spp $spp --max-path $mp --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-workload 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['roofline']['terms']; print('%-20s %-14s %8.1f Msamples/s' % ('$s', '$*', d['value']))"; }
b CornellBoxDiffuse 512 512 1024 8 A=1
b CornellBoxDiffuse 512 512 1024 8 A=2
b TestCaseFurnace 512 512 256 0 A=1
b TestCase0 512 512 256 0 A=1
python tools/bpt_prof.py CornellBoxDiffuse
