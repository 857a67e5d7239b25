#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 800 python tools/flat_limit.py 2>&1 | tee gpurun_out/r03_flat_limit.txt
