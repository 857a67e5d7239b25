#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_bptflat_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_bptflat_tests.log
for f in 0 1; do echo "MI_BPT_FLAT=$f"; for sc in CornellBoxDiffuse CornellBoxPhong TestCaseFurnace; do MI_BPT_FLAT=$f python tools/bpt_prof.py $sc 2>&1 | tail -1; done; done
