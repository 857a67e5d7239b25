#!/bin/bash
# A/B of the two hierarchy builders on the GPU box: bench.py per scene with MI_PT_BVH=lbvh|ploc.
#   tools/ab_bvh.sh > gpurun_out/ab_ploc.txt
set -e
run() {  # scene width height spp maxpath
  for b in lbvh ploc; do
    MI_PT_BVH=$b timeout -k 10 300 python bench.py --scene "$1" --width $2 --height $3 --spp $4 --max-path $5 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/ab_bvh.json 2> /tmp/ab_bvh.err
    python - "$1" $b <<'PY'
import json, sys
d = json.load(open("/tmp/ab_bvh.json")); t = d["roofline"]["terms"]
print("%-22s %-5s %8.0f Msamples/s  %8.2f ms  N %6.2f T %5.2f  N' %6.2f T' %5.2f  simd %.2f/%.2f" % (
    sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], t["N"], t["T"], t["N_shadow_per_segment"], t["T_shadow_per_segment"],
    t["simd_efficiency_closest_traversal"], t["simd_efficiency_shadow_traversal"]))
PY
  done
}
run CornellBoxDiffuse 512 512 1024 8
run CornellBoxSpecular 1024 1024 64 0
run MirrorBalls 960 540 64 0
run MetalRings 960 540 64 0
run LivingRoomLit 960 540 64 0
run atrium 960 540 64 0
run clutter 960 540 64 0
