#!/usr/bin/env python3
"""Image averages of the reference's TestCase models with BPT on the GPU (the reference's unit_test.py protocol)."""
import glob, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import master_amd as ma
names = sorted([os.path.basename(p)[:-8] for p in glob.glob(os.path.join(ROOT, "scenes", "TestCase*.miscene"))], key=lambda n: (len(n), n))
for n in names:
    s = ma.Scene.load(os.path.join(ROOT, "scenes", n + ".miscene"))
    pt = ma.PathTracing(s, beta=2.0)
    img = pt.bpt_render_rgbn(256, 256, spp=512, seed=1)
    rgb = img[..., :3] / np.maximum(img[..., 3:], 1)
    kinds = sorted(set(m.type for m in s.materials))
    print("%-16s tris %5d kinds %-16s mean %.4f %.4f %.4f  errors %d" % (n, s.n_triangles, kinds, rgb[..., 0].mean(), rgb[..., 1].mean(), rgb[..., 2].mean(), pt.last_stats.numeric_errors), flush=True)
