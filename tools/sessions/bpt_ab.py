#!/usr/bin/env python3
"""BPT render time with / without the visibility stage (MI_BPT_DYN_VIS): python tools/bpt_ab.py [scene ...]  (one process per setting: the env is read per launch)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import master_amd as ma
import numpy as np
for name in sys.argv[1:] or ["CornellBoxDiffuse", "LivingRoomLit", "CornellBoxSpecular"]:
    s = ma.Scene.load(os.path.join(ROOT, "scenes", name + ".miscene"))
    pt = ma.PathTracing(s, beta=2.0)
    out = {}
    for v in ("0", "1"):
        os.environ["MI_BPT_DYN_VIS"] = v
        pt.bpt_render_rgbn(512, 512, spp=4, seed=1)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); img = pt.bpt_render_rgbn(512, 512, spp=32, seed=1); dt = time.perf_counter() - t0
            best = min(best, dt)
        st = pt.last_stats
        out[v] = (best, st.num_basic_rays + st.num_shadow_rays, np.asarray(img, dtype=np.float64).sum())
    a, b = out["0"], out["1"]
    print("%-20s per-lane shadow rays %.1f ms (%.0f Mrays/s)   visibility stage %.1f ms (%.0f Mrays/s)   x%.2f   rays equal %s, image sums %.6g / %.6g" % (
        name, a[0] * 1e3, a[1] / a[0] / 1e6, b[0] * 1e3, b[1] / b[0] / 1e6, a[0] / b[0], a[1] == b[1], a[2], b[2]), flush=True)
