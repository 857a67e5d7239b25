#!/usr/bin/env python3
"""GPU check of the wavefront pipeline against the megakernel: per-path bit equality, image equality, throughput."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import master_amd as ma  # noqa: E402
from master_amd import scenegen  # noqa: E402

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def load(spec):
    p = os.path.join(ROOT, "scenes", spec + ".miscene")
    return ma.Scene.load(p) if os.path.exists(p) else scenegen.load(spec)


def main():
    for spec, W, H, spp in [("CornellBoxDiffuse", 256, 256, 16), ("CornellBoxSpecular", 256, 256, 16), ("MetalRings", 480, 270, 16), ("LivingRoomLit", 480, 270, 16), ("atrium:60000", 480, 270, 16)]:
        s = load(spec)
        a, b = ma.PathTracing(s), ma.PathTracing(s)
        a.set_kernel(ma.KERNEL_MEGA_GLOBAL); b.set_kernel(ma.KERNEL_WAVEFRONT)
        rng = np.random.default_rng(3); n = 20000
        xy = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.uint32); si = rng.integers(0, 64, n).astype(np.uint64)
        ra, ca = a.trace_paths(W, H, xy, si, seed=9); rb, cb = b.trace_paths(W, H, xy, si, seed=9)
        same = ((ra.view(np.uint32) == rb.view(np.uint32)) | (np.isnan(ra) & np.isnan(rb))).all()
        ia = a.render_rgbn(W, H, spp=spp, seed=5); sa = a.last_stats
        ib = b.render_rgbn(W, H, spp=spp, seed=5); sb = b.last_stats
        t0 = time.time(); ib2 = b.render_rgbn(W, H, spp=spp, seed=5); tb = time.time() - t0
        print("%-20s paths equal %s counts equal %s | image max|diff| %.3g denom equal %s | rays %d/%d shadow %d/%d err %d/%d | mega %.2f ms wf %.2f ms rounds %d deterministic %s" % (
            spec, same, np.array_equal(ca, cb), np.nanmax(np.abs(ia[..., :3] - ib[..., :3])), np.array_equal(ia[..., 3], ib[..., 3]),
            sa.num_basic_rays, sb.num_basic_rays, sa.num_shadow_rays, sb.num_shadow_rays, sa.numeric_errors, sb.numeric_errors,
            sa.trace_ms, sb.trace_ms, sb.wave_loop_bodies[0], np.array_equal(ib, ib2)))


if __name__ == "__main__":
    main()
