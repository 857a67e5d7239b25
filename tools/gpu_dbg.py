import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, master_amd as ma
s = ma.Scene.load(os.path.join(ROOT, "scenes", "CornellBoxDiffuse.miscene"))
pt = ma.PathTracing(s, max_path=8); pt.set_instrumented(True)
pt.render_rgbn(512, 512, spp=64, seed=1); st = pt.last_stats
lane_c = st.nodes_closest + st.tris_closest; lane_s = st.nodes_shadow + st.tris_shadow
print("segments %d  closest steps/lane-ray %.2f  shadow steps per shadow ray %.2f" % (st.num_basic_rays, lane_c / st.num_basic_rays, lane_s / max(1, st.num_shadow_rays)))
print("SIMD efficiency closest traversal: %.3f   shadow traversal: %.3f" % (lane_c / (64.0 * st.wave_steps_closest), lane_s / (64.0 * st.wave_steps_shadow)))
print("wave steps closest %d shadow %d" % (st.wave_steps_closest, st.wave_steps_shadow))
