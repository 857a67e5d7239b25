#!/usr/bin/env python3
"""Which scenes read from HBM gain from path regeneration in the BPT tracing stage?  Per scene: closest-hit rays per path, ms with MI_BPT_PERSIST=0 / 1 (512^2 x 16 frames)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import master_amd as ma
names = sorted(f[:-8] for f in os.listdir(os.path.join(ROOT, "scenes")) if f.endswith(".miscene"))
for name in names:
    s = ma.Scene.load(os.path.join(ROOT, "scenes", name + ".miscene"))
    if s.n_triangles <= 114: continue  # walked in LDS: one lane per path always
    out = []
    for v in ("0", "1", ""):
        if v: os.environ["MI_BPT_PERSIST"] = v
        else: os.environ.pop("MI_BPT_PERSIST", None)  # the default: by size, then by the rays per path measured on the handle's finished launches
        pt = ma.PathTracing(s, beta=2.0)
        pt.bpt_render_rgbn(512, 512, spp=16, seed=1)
        pt.bpt_render_rgbn(512, 512, spp=16, seed=1)
        st = pt.last_stats
        out.append(st.trace_ms)
        rpp = st.num_basic_rays / max(1, st.num_paths)
        del pt
    print("%-26s %6d triangles  %5.2f closest-hit rays per path   per lane %7.1f ms   regeneration %7.1f ms   %+5.1f %%   default %7.1f ms" % (name, s.n_triangles, rpp, out[0], out[1], 100.0 * (out[1] / out[0] - 1.0), out[2]), flush=True)
