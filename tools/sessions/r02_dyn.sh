#!/bin/bash
set -u
O=gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "per_path or render_equals or corpus or feature or numeric or full_size_workload" > $O/dyn_tests.log 2>&1; echo "pytest rc=$?"; tail -4 $O/dyn_tests.log
for d in 0 1; do MI_PT_DYN=$d timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-hbm-workload > $O/ab_dyn_$d.json 2>$O/ab_dyn.err; python -c "
import json; d=json.load(open('$O/ab_dyn_$d.json')); print('C2 dyn=$d %.1f Msamples/s  lds %d B' % (d['value'], d['config']['launch']['lds_bytes_per_workgroup']))"; done
for s in CornellBoxSpecular TestCaseFurnace; do for d in 0 1; do MI_PT_DYN=$d timeout -k 10 200 python bench.py --scene $s --spp 256 --max-path 0 --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-workload > $O/ab_dyn_${s}_$d.json 2>$O/ab_dyn.err; python -c "
import json; d=json.load(open('$O/ab_dyn_${s}_$d.json')); print('$s dyn=$d %.1f Msamples/s' % d['value'])"; done; done
