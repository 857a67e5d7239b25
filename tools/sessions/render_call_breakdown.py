#!/usr/bin/env python3
"""Where one Technique::render call of the host mirror spends its time (C2 scene, 512x512, 16 spp per call)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import master_amd as ma
s = ma.Scene.load(os.path.join(ROOT, "scenes", "CornellBoxDiffuse.miscene"))
pt = ma.PathTracing(s, max_path=8)
w = h = 512
view = np.zeros((h, w, 4), np.float64); ref = np.full((h, w, 3), 0.5, np.float32)
for spp in (1, 16, 64):
    pt.render_rgbn(w, h, spp=spp, seed=1)
    t = {"render_rgbn": 0.0, "kernel (device events)": 0.0, "view_add_frame": 0.0, "rms_abs_errors_view": 0.0}
    n = 20
    for k in range(n):
        t0 = time.perf_counter(); rgbn = pt.render_rgbn(w, h, spp=spp, seed=1, sample_offset=k * spp); t1 = time.perf_counter()
        ma.view_add_frame(view, rgbn); t2 = time.perf_counter()
        ma.rms_abs_errors_view(view, ref); t3 = time.perf_counter()
        t["render_rgbn"] += t1 - t0; t["kernel (device events)"] += pt.last_stats.trace_ms * 1e-3; t["view_add_frame"] += t2 - t1; t["rms_abs_errors_view"] += t3 - t2
    print("spp %3d per call: " % spp + ", ".join("%s %.3f ms" % (k, v / n * 1e3) for k, v in t.items()), flush=True)
