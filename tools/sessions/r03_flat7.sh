#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_flat7_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r03_flat7_tests.log
