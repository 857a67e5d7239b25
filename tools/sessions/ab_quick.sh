#!/bin/bash
# quick throughput table of the current build on the GPU box:  tools/ab_quick.sh [label]
run() {
  timeout -k 10 300 python bench.py --scene "$1" --width $2 --height $3 --spp $4 --max-path $5 --steps 2 --warmup 1 --no-cpu-baseline $EXTRA > /tmp/abq.json 2> /tmp/abq.err || { echo "$1 FAILED"; tail -3 /tmp/abq.err; return; }
  python - "$1" "$LABEL" <<'PY'
import json, sys
d = json.load(open("/tmp/abq.json")); t = d["roofline"]["terms"]
print("%-8s %-22s %8.0f Msamples/s  %8.2f ms  N %6.2f T %5.2f  N' %6.2f T' %5.2f  simd %.2f/%.2f" % (
    sys.argv[2], sys.argv[1], d["value"], d["ms_per_step"], t["N"], t["T"], t["N_shadow_per_segment"], t["T_shadow_per_segment"],
    t["simd_efficiency_closest_traversal"] or 0.0, t["simd_efficiency_shadow_traversal"] or 0.0))
PY
}
LABEL=${1:-head}
EXTRA=${2:-}
run CornellBoxDiffuse 512 512 1024 8
run CornellBoxSpecular 1024 1024 64 0
run MetalRings 960 540 64 0
run LivingRoomLit 960 540 64 0
run atrium 960 540 64 0
run clutter 960 540 64 0
