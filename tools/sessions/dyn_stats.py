"""Loop-body census of the unified dynamic-fetch traversal (diagnostic build -DMI_DYN_STATS) on C2."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import master_amd as ma
scene = ma.Scene.load(os.path.join(ROOT, "scenes", (sys.argv[1] if len(sys.argv) > 1 else "CornellBoxDiffuse") + ".miscene"))
pt = ma.PathTracing(scene, max_path=8)
fb = torch.zeros((512, 512, 4), dtype=torch.float32, device="cuda")
pt.render_device(fb.data_ptr(), 512, 512, spp=64, seed=1)
st = pt.render_device(fb.data_ptr(), 512, 512, spp=64, seed=1)
c = list(st.phase_cycles)
trips = max(c[5], 1)
print("trace_ms %.3f  Msamples/s %.1f" % (st.trace_ms, st.num_basic_rays / st.trace_ms / 1e3))
print("per trip: iterations %.1f  node bodies %.1f  leaf bodies %.1f  refills %.2f  rays fetched %.1f   (trips %d, segments/trip %.1f, shadow/trip %.1f)" % (
    c[0] / trips, c[1] / trips, c[2] / trips, c[3] / trips, c[4] / trips, trips, st.num_basic_rays / trips, st.num_shadow_rays / trips))
