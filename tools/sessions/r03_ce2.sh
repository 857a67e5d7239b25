#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_ce2_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_ce2_tests.log
timeout -k 10 600 python tools/flat_limit.py 2>&1 | grep -v "^soup\|TestCase1[0-9]\|TestCase2\|TestCase3\|TestCase[5-7] " | tee gpurun_out/r03_ce2_limit.txt
python tools/bpt_prof.py CornellBoxDiffuse 2>&1 | tail -1
python tools/bpt_prof.py LivingRoomLit 2>&1 | tail -1
python tools/bpt_prof.py MetalRings 2>&1 | tail -1
