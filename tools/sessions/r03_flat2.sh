#!/bin/bash
# r03: first contact of the flat leaf list: parity tests, then C2 A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r03_flat2_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r03_flat2_tests.log
tail -3 gpurun_out/r03_flat2_tests.log
for f in 0 1; do
  MI_PT_FLAT=$f timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hbm-workload > gpurun_out/r03_flat2_bench_$f.json 2> gpurun_out/r03_flat2_bench_$f.err
  python -c "import json; d=json.load(open('gpurun_out/r03_flat2_bench_$f.json')); print('FLAT=$f', round(d['value']), d["ms_per_step"], d["config"].get("launch"))"
done
