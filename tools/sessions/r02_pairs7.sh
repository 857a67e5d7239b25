#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_cadence_c5.py -m gpu -x -q -k "pair_leaves or dynamic_fetch" > $O/pairs_tests2.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pairs_tests2.log
[ $rc = 0 ] || exit 1
for sc in LivingRoomLit CornellBoxSpecular; do
  echo "$sc PRE waves 4:"; python tools/bpt_prof.py $sc
  echo "$sc PRE waves 6:"; MI_PT_LIB=$PWD/master_amd/libmi_pt_pre6.so python tools/bpt_prof.py $sc
done
