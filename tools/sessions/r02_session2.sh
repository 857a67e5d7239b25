#!/bin/bash
set -u
O=gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/gpu_tests.log
timeout -k 10 300 python bench.py --steps 5 --warmup 1 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02/bench_n1.json"))
print("C2 %.1f Msamples/s, %.2f ms/step; hbm_workload %.1f Msamples/s %.1f ms/step tables_in_lds=%s" % (d["value"], d["ms_per_step"], d["hbm_workload"]["value"], d["hbm_workload"]["ms_per_step"], d["hbm_workload"]["tables_in_lds"]))
PY
for s in LivingRoomLit MetalRings; do
  for t in 0 1; do MI_PT_LDS_TABLES=$t timeout -k 10 200 python bench.py --scene $s --width 1920 --height 1080 --spp 128 --max-path 0 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload > $O/ab_tables_${s}_$t.json 2>$O/ab_tables.err; python -c "
import json; d=json.load(open('$O/ab_tables_${s}_$t.json')); print('$s tables=$t %.1f Msamples/s' % d['value'])"; done; done
for t in 0 1; do MI_PT_LDS_TABLES=$t timeout -k 10 200 python bench.py --scene atrium --width 1920 --height 1080 --spp 64 --max-path 0 --steps 2 --warmup 1 --no-cpu-baseline --no-hbm-workload > $O/ab_tables_atrium_$t.json 2>$O/ab_tables.err; python -c "
import json; d=json.load(open('$O/ab_tables_atrium_$t.json')); print('atrium tables=$t %.1f Msamples/s' % d['value'])"; done
