#!/usr/bin/env python3
"""Where does the LDS-resident megakernel stop paying?  Seeded soups of growing size, both kernels (GPU box)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import master_amd as ma
from master_amd import scenegen as sb
os.environ.setdefault("MI_PT_LDS_LIMIT_KB", "156")  # let the LDS kernel run beyond the product's 48 KB rule: the point is to find where it stops paying
for n in (30, 70, 90, 110, 140, 180, 240, 320):
    s = sb.random_soup(n, seed=3)
    row = []
    for k in (ma.KERNEL_MEGA_LDS, ma.KERNEL_MEGA_GLOBAL):
        pt = ma.PathTracing(s, max_path=8)
        try:
            pt.set_kernel(k)
        except ma.MiError:
            row.append(None); continue
        pt.render_rgbn(512, 512, spp=16, seed=1)
        best = 1e9
        for _ in range(2):
            pt.render_rgbn(512, 512, spp=256, seed=1); st = pt.last_stats
            best = min(best, st.trace_ms)
        row.append(st.num_basic_rays / best / 1e3)
    info = ma.PathTracing(s).bvh_info()
    print("soup %3d tris (depth %2d): LDS %s  HBM %s Msamples/s" % (s.indices.shape[0], info.max_depth, "%.0f" % row[0] if row[0] else "n/a", "%.0f" % row[1]), flush=True)
