#!/usr/bin/env python3
"""Dump the rays on which the flat hooks and the oracle disagree (GPU box) -> gpurun_out/flat_debug.npz"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import master_amd as ma, oracle
from conftest import load_scene
name = sys.argv[1] if len(sys.argv) > 1 else "CornellBoxPhong"
s = load_scene(name)
os.environ["MI_PT_INTERSECT_FLAT"] = "1"
pt, orc = ma.PathTracing(s), oracle.Oracle(s)
rng = np.random.default_rng(9)
n = 60000
lo, hi = s.positions.min(0), s.positions.max(0)
o = np.zeros(n, ma.SURFACE_DTYPE)
o["position"] = rng.uniform(lo, hi, (n, 3)); g = rng.normal(size=(n, 3)); o["gnormal"] = g / np.linalg.norm(g, axis=1, keepdims=True)
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True); d = d.astype(np.float32)
k = n // 6
d[:k, rng.integers(0, 3, k)] = 0.0
ax = rng.integers(0, 3, k); d[k:2 * k] = 0.0; d[np.arange(k, 2 * k), ax] = rng.choice([-1.0, 1.0], k)
tri = s.indices[rng.integers(0, len(s.indices), k)]
o["position"][2 * k:3 * k] = s.positions[tri[:, 0]]
tgt = s.positions[s.indices[rng.integers(0, len(s.indices), k), rng.integers(0, 3, k)]]
dv = tgt - s.positions[tri[:, 0]]; nz = np.linalg.norm(dv, axis=1) > 0
d[2 * k:3 * k][nz] = (dv[nz] / np.linalg.norm(dv[nz], axis=1, keepdims=True)).astype(np.float32)
w = rng.uniform(0, 1, (k, 3)); w /= w.sum(1, keepdims=True)
tri2 = s.indices[rng.integers(0, len(s.indices), k)]
p = (s.positions[tri2] * w[:, :, None]).sum(1)
o["position"][3 * k:4 * k] = p
e = s.positions[tri2[:, 1]] - s.positions[tri2[:, 0]]; en = np.linalg.norm(e, axis=1, keepdims=True); en[en == 0] = 1
d[3 * k:4 * k] = (e / en).astype(np.float32)
o["position"][4 * k:5 * k] = np.where(rng.integers(0, 2, (k, 3)) == 0, lo, hi)
tg = np.zeros(n, ma.SURFACE_DTYPE)
tg["position"] = np.roll(o["position"], 17, axis=0); tg["gnormal"] = np.roll(o["gnormal"], 5, axis=0)
gv, ov = pt.occluded(o, tg), orc.occluded(o, tg)
os.environ["MI_PT_INTERSECT_FLAT"] = "0"
tv = pt.occluded(o, tg)
bad = np.nonzero(gv != ov)[0]
print("mismatches flat vs oracle:", len(bad), "tree vs oracle:", int((tv != ov).sum()), "first:", bad[:10], "flat", gv[bad[:10]], "oracle", ov[bad[:10]])
np.savez(os.path.join(ROOT, "gpurun_out", "flat_debug.npz"), bad=bad, o_pos=o["position"][bad], o_gn=o["gnormal"][bad], t_pos=tg["position"][bad], t_gn=tg["gnormal"][bad], flat=gv[bad], orc=ov[bad])
