#!/bin/bash
set -u
O=gpurun_out/r02; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "per_path or render_equals or corpus or intersect" > $O/b128_tests.log 2>&1; echo "pytest rc=$?"; tail -2 $O/b128_tests.log
b() { env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-hbm-workload 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-60s %.1f Msamples/s' % ('$*', d['value']))"; }
b MI_PT_DYN=0
b MI_PT_DYN=0 MI_PT_LIB=$PWD/master_amd/libmi_pt_nob128.so
b MI_PT_DYN=1
b MI_PT_DYN=1 MI_PT_LIB=$PWD/master_amd/libmi_pt_nob128.so
b MI_PT_DYN=0
b MI_PT_DYN=0 MI_PT_LIB=$PWD/master_amd/libmi_pt_nob128.so
