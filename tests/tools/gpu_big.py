#!/usr/bin/env python3
"""Large-scene check on a GPU box: parity of a sample of paths vs the oracle + throughput + traversal stats."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, master_amd as ma, oracle
from master_amd import scenegen as procedural
spec = sys.argv[1] if len(sys.argv) > 1 else "atrium"
W, H, spp = (int(x) for x in (sys.argv[2:5] if len(sys.argv) > 4 else (960, 540, 16)))
s = procedural.load(spec) if spec.split(":")[0] in procedural.SCENES else ma.Scene.load(os.path.join(ROOT, "scenes", spec + ".miscene"))
t = time.time(); pt = ma.PathTracing(s); info = pt.bvh_info()
print("%s: %d tris, create %.2fs, LBVH build %.2f ms, depth %d, stack %d, kernel %d" % (spec, s.n_triangles, time.time() - t, info.build_ms, info.max_depth, info.stack_entries, pt.get_kernel()))
orc = oracle.Oracle(s)
gn, gs, gm = pt.bvh(); on, os_, om = orc.bvh()
print("LBVH bit-exact:", np.array_equal(gm, om), np.array_equal(gs, os_), gn.tobytes() == on.tobytes())
rng = np.random.default_rng(0); n = 20000
xy = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.uint32); si = rng.integers(0, 1000, n).astype(np.uint64)
g, gc = pt.trace_paths(W, H, xy, si, seed=3); o, oc = orc.trace_paths(W, H, xy, si, seed=3)
same = ((g.view(np.uint32) == o.view(np.uint32)) | (np.isnan(g) & np.isnan(o))).all(1)
close = np.isclose(g, o, rtol=2e-5, atol=1e-7, equal_nan=True).all(1)
print("paths: exact %.5f close %.5f counts equal %.5f  (Phong materials use library powf: close is the bar there)" % (same.mean(), close.mean(), (gc == oc).all(1).mean()))
pt.render_rgbn(W, H, spp=2, seed=1)
img = pt.render_rgbn(W, H, spp=spp, seed=1); st = pt.last_stats
print("render %dx%dx%d: %.1f ms, %.1f Msamples/s, %.1f Mrays/s, Lbar %.2f, numeric errors %d" % (W, H, spp, st.trace_ms, st.num_basic_rays / st.trace_ms / 1e3, (st.num_basic_rays + st.num_shadow_rays) / st.trace_ms / 1e3, st.num_basic_rays / st.num_paths, st.numeric_errors))
pt.set_instrumented(True); pt.render_rgbn(W, H, spp=max(1, spp // 4), seed=1); st = pt.last_stats
seg = st.num_basic_rays
print("per segment: N %.1f T %.1f | shadow N' %.1f T' %.1f (s=%.2f) | SIMD eff closest %.3f shadow %.3f" % (st.nodes_closest / seg, st.tris_closest / seg, st.nodes_shadow / max(1, st.num_shadow_rays), st.tris_shadow / max(1, st.num_shadow_rays), st.num_shadow_rays / seg,
      (st.nodes_closest + st.tris_closest) / (64.0 * st.wave_steps_closest), (st.nodes_shadow + st.tris_shadow) / (64.0 * max(1, st.wave_steps_shadow))))
