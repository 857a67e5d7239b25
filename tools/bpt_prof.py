#!/usr/bin/env python3
"""One BPT render of CornellBoxDiffuse 512^2 x 64 for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import master_amd as ma
s = ma.Scene.load(os.path.join(ROOT, "scenes", (sys.argv[1] if len(sys.argv) > 1 else "CornellBoxDiffuse") + ".miscene"))
pt = ma.PathTracing(s, beta=2.0)
pt.bpt_render_rgbn(512, 512, spp=64, seed=1)
pt.bpt_render_rgbn(512, 512, spp=64, seed=1)
st = pt.last_stats
print("%.1f ms, %.0f Mrays/s" % (st.trace_ms, (st.num_basic_rays + st.num_shadow_rays) / st.trace_ms / 1e3))
