#!/bin/bash
# r03: centre-form box loop; SLP vectoriser on/off; parity tests with the new loop
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r03_flat4_tests.log 2>&1; echo "tests rc=$?"
tail -2 gpurun_out/r03_flat4_tests.log
for lib in libmi_pt.so libmi_pt_noslp.so; do
  for f in 0 1; do
    MI_PT_LIB=$PWD/master_amd/$lib MI_PT_FLAT=$f timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hbm-workload > gpurun_out/r03_flat4_$lib.$f.json 2> gpurun_out/r03_flat4_$lib.$f.err
    python -c "import json; d=json.load(open('gpurun_out/r03_flat4_$lib.$f.json')); print('$lib FLAT=$f', round(d['value']), round(d['ms_per_step'],2), d['config'].get('launch'))"
  done
done
