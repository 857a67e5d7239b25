#!/usr/bin/env python3
"""mi_pt_render_multi at the C5 shape (3840x2160): frame put together on the first device (r03) against on the host (r02), N handles.
One GPU per box, so the handles share it and 'peer reads' are local — what this shows is the PCIe side: N framebuffers to the host against one."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import master_amd as ma
from master_amd import scenegen as sb
s = sb.load("clutter")
W, H = 3840, 2160
for n in (2, 4, 8):
    pts = [ma.PathTracing(s, max_path=4) for _ in range(n)]
    row = []
    for merge in ("device", "host"):
        if merge == "host": os.environ["MI_PT_MULTI_HOST_MERGE"] = "1"
        else: os.environ.pop("MI_PT_MULTI_HOST_MERGE", None)
        ma.render_multi(pts, W, H, spp=1, seed=1)
        t0 = time.perf_counter()
        for k in range(3):
            img, st = ma.render_multi(pts, W, H, spp=1, seed=1, sample_offset=1 + k)
        dt = (time.perf_counter() - t0) / 3
        assert ma.lib().mi_pt_last_multi_merge() == (1 if merge == "device" else 0)
        row.append((merge, dt * 1e3, st.gpu_ms, img))
    assert np.array_equal(row[0][3].view(np.uint32), row[1][3].view(np.uint32))
    print("%d handles, %dx%d, 1 spp: device-side merge %.1f ms per call, host-side merge %.1f ms (identical bits); kernels %.1f ms" % (n, W, H, row[0][1], row[1][1], row[0][2]), flush=True)
    del pts
