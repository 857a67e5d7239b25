#!/usr/bin/env python3
"""Wide randomised parity sweep (GPU box): every corpus fixture x parameter sets x both estimators, many more paths than the
unit tests use.  Prints one line per (scene, technique, parameters) and a summary; exit code 1 on any mismatch.
    python tests/tools/parity_sweep.py [paths_per_case]
"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import master_amd as ma
import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
scenes = sorted(f[:-8] for f in os.listdir(os.path.join(ROOT, "scenes")) if f.endswith(".miscene") and os.path.getsize(os.path.join(ROOT, "scenes", f)) < 1000000)
params = [dict(beta=1.0, roulette=0.9, max_path=ma.PTRDIFF_MAX), dict(beta=2.0, roulette=0.7, max_path=6), dict(beta=1.5, roulette=0.95, max_path=ma.PTRDIFF_MAX),
          dict(beta=0.0, roulette=0.5, max_path=3)]
bits = lambda a, b: ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)))
bad = 0; cases = 0; paths = 0; t0 = time.time()
for name in scenes:
    s = ma.Scene.load(os.path.join(ROOT, "scenes", name + ".miscene"))
    for k, p in enumerate(params):
        rng = np.random.default_rng(1000 + k)
        W, H = 160, 90
        xy = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.uint32); si = rng.integers(0, 1 << 20, n).astype(np.uint64)
        pt = ma.PathTracing(s, **p); orc = oracle.Oracle(s, **p)
        g, gc = pt.trace_paths(W, H, xy, si, seed=77 + k); r, rc = orc.trace_paths(W, H, xy, si, seed=77 + k)
        m_pt = int((~bits(g, r).all(1)).sum()) + int((gc != rc).any(1).sum())
        m_bpt = 0
        if p["max_path"] == ma.PTRDIFF_MAX or True:
            pb = ma.PathTracing(s, beta=p["beta"], roulette=p["roulette"]); ob = oracle.Oracle(s, beta=p["beta"], roulette=p["roulette"])
            nb = n // 4
            gr, gs, gcb = pb.bpt_trace_paths(W, H, xy[:nb], si[:nb], seed=5 + k); orr, os_, ocb = ob.bpt_trace_paths(W, H, xy[:nb], si[:nb], seed=5 + k)
            m_bpt = int((~bits(gr, orr).all(1)).sum()) + int((~bits(gs, os_).all(1)).sum()) + int((gcb != ocb).any(1).sum())
            paths += nb
        cases += 2; paths += n; bad += (m_pt > 0) + (m_bpt > 0)
        print("%-22s beta %.1f roulette %.2f max_path %-4s PT mismatches %d   BPT mismatches %d" % (
            name, p["beta"], p["roulette"], "inf" if p["max_path"] == ma.PTRDIFF_MAX else p["max_path"], m_pt, m_bpt), flush=True)
# image mode runs the feature-specialised megakernel variants (the list mode above runs the general one): frames against the oracle
img_bad = 0
for name in scenes:
    s = ma.Scene.load(os.path.join(ROOT, "scenes", name + ".miscene"))
    for k, p in enumerate(params[:3]):
        pt = ma.PathTracing(s, **p); orc = oracle.Oracle(s, **p)
        a = pt.render_rgbn(96, 54, spp=6, seed=31 + k, sample_offset=9); st = pt.last_stats
        b = orc.render_rgbn(96, 54, spp=6, seed=31 + k, sample_offset=9); so = orc.last_stats
        same = (np.isclose(a, b, rtol=1.2e-7, atol=0) | (np.isnan(a) & np.isnan(b))).all() and np.array_equal(a[..., 3], b[..., 3])
        counts = (st.num_basic_rays, st.num_shadow_rays, st.numeric_errors) == (so.num_basic_rays, so.num_shadow_rays, so.numeric_errors)
        cases += 1
        if not (same and counts):
            img_bad += 1; bad += 1
            print("IMAGE MISMATCH %s params %d: pixels %s counts %s" % (name, k, same, counts), flush=True)
print("images: %d scenes x 3 parameter sets, %d mismatching frames" % (len(scenes), img_bad))
# BPT frames through mi_bpt_render (r04: regenerating tracing kernels on the large models, two launches in flight, long paths set aside): 262 144 paths per scene —
# enough for the launches in flight to engage — against the oracle's frames; the light-image splats and a pixel's frames add up in free FP64 order
bpt_bad = 0
for name in scenes:
    s = ma.Scene.load(os.path.join(ROOT, "scenes", name + ".miscene"))
    pb = ma.PathTracing(s, beta=2.0); ob = oracle.Oracle(s, beta=2.0)
    a = pb.bpt_render_rgbn(256, 128, spp=8, seed=3); st = pb.last_stats
    b = ob.bpt_render_rgbn(256, 128, spp=8, seed=3); so = ob.last_stats
    same = (np.isclose(a, b, rtol=2e-6, atol=1e-12) | (np.isnan(a) & np.isnan(b))).all() and np.array_equal(a[..., 3], b[..., 3])
    counts = (st.num_basic_rays, st.num_shadow_rays, st.numeric_errors) == (so.num_basic_rays, so.num_shadow_rays, so.numeric_errors)
    cases += 1; paths += 256 * 128 * 8
    if not (same and counts):
        bpt_bad += 1; bad += 1
        print("BPT IMAGE MISMATCH %s: pixels %s counts %s" % (name, same, counts), flush=True)
print("BPT images: %d scenes, %d mismatching" % (len(scenes), bpt_bad))
print("SUMMARY: %d scenes, %d cases, %d paths, %d cases with mismatches, %.0f s" % (len(scenes), cases, paths, bad, time.time() - t0))
sys.exit(1 if bad else 0)
