#!/bin/bash
for lib in libmi_pt.so; do
  for sc in CornellBoxDiffuse LivingRoomLit; do
  [ $lib = libmi_ab_b0.so ] && [ $sc = LivingRoomLit ] && continue
  MI_PT_LIB=$GRAFT_REPO_ROOT/master_amd/$lib SCENE=$sc python - <<PY
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import master_amd as ma
s = ma.Scene.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "scenes", os.environ["SCENE"] + ".miscene"))
pt = ma.PathTracing(s, beta=2.0)
pt.bpt_render_rgbn(512, 512, spp=2, seed=1)
best = None
for k in range(2):
    pt.bpt_render_rgbn(512, 512, spp=32, seed=1); st = pt.last_stats
    best = st.trace_ms if best is None or st.trace_ms < best else best
print("%-16s %-20s %.1f ms, %.1f Mrays/s" % (os.path.basename(os.environ["MI_PT_LIB"]), os.environ["SCENE"], best, (st.num_basic_rays + st.num_shadow_rays) / best / 1e3), flush=True)
PY
  done
done
