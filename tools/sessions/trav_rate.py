#!/usr/bin/env python3
"""Pure traversal kernel rate (k_intersect) for coherent / shuffled / random rays — run under rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import master_amd as ma
from master_amd import scenegen

spec = sys.argv[1] if len(sys.argv) > 1 else "atrium"
s = scenegen.load(spec) if spec.split(":")[0] in scenegen.SCENES else ma.Scene.load(os.path.join(os.path.dirname(__file__), "..", "scenes", spec + ".miscene"))
pt = ma.PathTracing(s)
W, H = 1920, 1080
cam = s.cameras[0]
fr = ma.camera_setup(cam, W / H)
v2w = np.array(list(fr.view_to_world), np.float32).reshape(3, 3).T  # columns
n = W * H
# tile order (8x8) like the megakernel
ty, tx, py, px = np.meshgrid(np.arange(H // 8), np.arange(W // 8), np.arange(8), np.arange(8), indexing="ij")
X = (tx * 8 + px).ravel().astype(np.float32) + 0.5; Y = (ty * 8 + py).ravel().astype(np.float32) + 0.5
d = np.stack([2 * X / H - W / H, 2 * Y / H - 1, np.full_like(X, -fr.focal_length_y)], 1)
d /= np.linalg.norm(d, axis=1, keepdims=True)
dw = (d @ v2w.T).astype(np.float32)
o = np.zeros(len(dw), ma.SURFACE_DTYPE); o["position"] = np.array(list(cam.position), np.float32); o["gnormal"] = -v2w[:, 2]
import time
def run(label, o, dw):
    t = time.time(); h, tt, p = pt.intersect(o, dw); dt = time.time() - t
    print(label, len(dw), "rays, hit fraction %.3f" % (p != ma.UINT32_MAX).mean(), "wall %.3f s" % dt, flush=True)
    return h, tt, p
h, tt, p = run("primary, tile order   ", o, dw)
perm = np.random.default_rng(1).permutation(len(dw))
run("primary, shuffled     ", o[perm], dw[perm])
# secondary rays: from the primary hits, cosine-ish random directions
ok = p != ma.UINT32_MAX
o2 = h[ok].copy(); rng = np.random.default_rng(2)
d2 = rng.normal(size=(ok.sum(), 3)).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
flip = (d2 * o2["gnormal"]).sum(1) < 0; d2[flip] *= -1
run("secondary, tile order ", o2, d2)
perm = rng.permutation(len(d2))
run("secondary, shuffled   ", o2[perm], d2[perm])

# sorted secondary rays: key = direction octant (3 bits) | 30-bit Morton code of the origin
def morton30(p):
    lo, hi = p.min(0), p.max(0)
    q = np.clip(((p - lo) / (hi - lo + 1e-20) * 1024).astype(np.uint64), 0, 1023)
    def ex(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    return (ex(q[:, 0]) << 2) | (ex(q[:, 1]) << 1) | ex(q[:, 2])
m = morton30(o2["position"].astype(np.float64))
octant = ((d2[:, 0] < 0).astype(np.uint64) << 2) | ((d2[:, 1] < 0).astype(np.uint64) << 1) | (d2[:, 2] < 0).astype(np.uint64)
for label, key in (("secondary, sorted by origin Morton          ", m), ("secondary, sorted by octant | origin Morton ", (octant << 30) | m),
                   ("secondary, sorted by origin Morton>>12 | octant | low", ((m >> 12) << 15) | (octant << 12) | (m & 0xFFF))):
    order = np.argsort(key, kind="stable")
    run(label, o2[order], d2[order])
