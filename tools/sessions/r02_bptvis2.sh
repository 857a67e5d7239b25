#!/bin/bash
# BPT visibility stage A/B: refill threshold, node records
O=gpurun_out/r02; mkdir -p $O
: > $O/bptvis_ab2.txt
for cfg in "16 0" "8 0" "32 0" "16 1"; do
  set -- $cfg
  echo "== TH $1 wide $2" >> $O/bptvis_ab2.txt
  MI_BPT_VIS_TH=$1 MI_BPT_VIS_WIDE=$2 timeout -k 10 300 python tools/bpt_ab.py CornellBoxDiffuse LivingRoomLit CornellBoxSpecular MetalRings >> $O/bptvis_ab2.txt 2>&1 || exit 1
done
cat $O/bptvis_ab2.txt
