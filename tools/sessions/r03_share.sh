#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_share_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_share_tests.log
bash tools/ab_libs.sh r03_share libmi_pt.so
python bench.py --scene TestCaseFurnace --width 512 --height 512 --spp 512 --max-path 8 --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('furnace', round(d['value']))"
python bench.py --scene CornellBoxPhong --width 512 --height 512 --spp 512 --max-path 8 --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-workload --no-time-to-rmse 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('phong', round(d['value']))"
