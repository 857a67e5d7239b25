#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
b() { local s=$1 spp=$2 mp=$3; shift 3; env "$@" timeout -k 10 300 python bench.py --scene $s --spp $spp --max-path $mp --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-workload $K 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['roofline']['terms']; print('%-20s %-34s %-12s %8.1f Msamples/s  lds %5d  nodes %s dyn-eff %s' % ('$s', '$*', '$K', d['value'], d['config']['launch']['lds_bytes_per_workgroup'], d['config']['kernel'], t.get('simd_efficiency_unified_traversal')))"; }
for s in "CornellBoxDiffuse 1024 8" "TestCaseFurnace 256 0" "TestCase0 256 0" "DoubleLight 256 0"; do
  K="" b $s A=auto
  K="--kernel 2" b $s A=hbm
  K="--kernel 2" b $s MI_PT_WIDE_NODES=0
  K="--kernel 2" b $s MI_PT_DYN=0
done
