#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_bpt.py -m gpu -x -q > $O/bptev_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/bptev_tests.log
[ $rc = 0 ] || exit 1
bash tools/bpt_prof.sh CornellBoxDiffuse LivingRoomLit MetalRings CornellBoxSpecular > $O/bptev_prof.txt 2>&1; grep "^==" $O/bptev_prof.txt
