for lib in libmi_bw2.so libmi_bw3.so libmi_pt.so libmi_bw5.so libmi_bw6.so; do
  MI_PT_LIB=$GRAFT_REPO_ROOT/master_amd/$lib python - <<PY
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import master_amd as ma
s = ma.Scene.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "scenes", "CornellBoxDiffuse.miscene"))
pt = ma.PathTracing(s, beta=2.0)
pt.bpt_render_rgbn(512, 512, spp=2, seed=1)
pt.bpt_render_rgbn(512, 512, spp=32, seed=1); st = pt.last_stats
print("$lib", "%.1f ms, %.1f Mrays/s" % (st.trace_ms, (st.num_basic_rays + st.num_shadow_rays) / st.trace_ms / 1e3))
PY
done
