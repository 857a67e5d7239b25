#!/usr/bin/env python3
"""Flat leaf list against the tree walk on LDS-resident scenes of growing leaf count (GPU box): where does the uniform box loop stop paying?"""
import glob, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import master_amd as ma
from master_amd import scenegen as sb


def rate(s, flat, mp=8):
    os.environ["MI_PT_FLAT"] = str(flat)
    pt = ma.PathTracing(s, max_path=mp)
    pt.render_rgbn(512, 512, spp=16, seed=1)
    best = 1e9
    for _ in range(2):
        pt.render_rgbn(512, 512, spp=256, seed=1); st = pt.last_stats
        best = min(best, st.trace_ms)
    return st.num_basic_rays / best / 1e3, pt.last_launch().flat_leaves


rows = []
for n in (6, 10, 14, 18, 22, 26, 30, 32):
    rows.append(("soup %d" % n, sb.random_soup(n, seed=3)))
for f in sorted(glob.glob(os.path.join(ROOT, "scenes", "*.miscene"))):
    s = ma.Scene.load(f)
    if s.indices.shape[0] <= 64:
        rows.append((os.path.basename(f)[:-8], s))
for name, s in rows:
    t, _ = rate(s, 0)
    f, k = rate(s, 1)
    print("%-28s %3d tris  %2d leaves  tree %7.0f  flat %7.0f Msamples/s  %+5.1f %%" % (name, s.indices.shape[0], k, t, f, 100.0 * (f / t - 1.0)) if k else
          "%-28s %3d tris  (no flat table)  tree %7.0f" % (name, s.indices.shape[0], t), flush=True)
