import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import master_amd as ma
for name in sys.argv[1:]:
    s = ma.Scene.load(os.path.join(ROOT, "scenes", name + ".miscene"))
    pt = ma.PathTracing(s, beta=2.0)
    pt.bpt_render_rgbn(512, 512, spp=64, seed=1)  # the same call once before it is timed: every buffer at its working size (an allocation that follows the
    # destruction of the previous model's 116 GB arena waits seconds for the driver, profiles/r04/ab_bpt_steps.txt #5)
    t0 = time.perf_counter()
    pt.bpt_render_rgbn(512, 512, spp=64, seed=1)
    dt = time.perf_counter() - t0
    st = pt.last_stats
    print("%s steps=%s: %.1f ms device, %.1f ms wall, %.0f Mrays/s" % (name, os.environ.get("MI_BPT_STEPS", "default"), st.trace_ms, dt * 1e3, (st.num_basic_rays + st.num_shadow_rays) / st.trace_ms / 1e3), flush=True)
