#!/bin/bash
# BPT visibility stage: parity suite, then A/B timings (forced off / on) and the default
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_bpt.py -x -q -m gpu > $O/bptvis_tests.log 2>&1; rc=$?
tail -5 $O/bptvis_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/bpt_ab.py CornellBoxDiffuse LivingRoomLit CornellBoxSpecular MetalRings > $O/bptvis_ab.txt 2>&1
cat $O/bptvis_ab.txt
timeout -k 10 300 python tools/bpt_prof.py LivingRoomLit; timeout -k 10 300 python tools/bpt_prof.py MetalRings; timeout -k 10 300 python tools/bpt_prof.py CornellBoxDiffuse
