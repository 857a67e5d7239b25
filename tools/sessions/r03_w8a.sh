#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_cadence_c5.py -m gpu -x -q -k "wide8 or dynamic_fetch" > gpurun_out/r03_w8a_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03_w8a_tests.log
bash tools/sessions/r03_spill1.sh
