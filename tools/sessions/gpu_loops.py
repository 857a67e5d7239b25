import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import master_amd as ma
from master_amd import scenegen
for spec, (W, H, spp, mp) in (("CornellBoxDiffuse", (512, 512, 64, 8)), ("CornellBoxSpecular", (512, 512, 16, ma.PTRDIFF_MAX)), ("atrium", (960, 540, 4, ma.PTRDIFF_MAX))):
    s = scenegen.load(spec) if spec in scenegen.SCENES else ma.Scene.load(os.path.join(ROOT, "scenes", spec + ".miscene"))
    pt = ma.PathTracing(s, max_path=mp); pt.set_instrumented(True)
    pt.render_rgbn(W, H, spp=spp, seed=1); st = pt.last_stats
    trips = st.num_basic_rays / 64.0  # lower bound on wave trips (full waves)
    b = list(st.wave_loop_bodies)
    print("%-20s per lane-ray: closest %.1f nodes %.1f tris | per wave trip (>= segments/64): node bodies %.1f leaf bodies %.1f | shadow: node bodies %.1f leaf bodies %.1f | slowest-lane steps closest %.1f shadow %.1f" % (
        spec, st.nodes_closest / st.num_basic_rays, st.tris_closest / st.num_basic_rays, b[0] / trips, b[1] / trips, b[2] / trips, b[3] / trips,
        st.wave_steps_closest / trips, st.wave_steps_shadow / trips))
