#!/usr/bin/env python3
"""Flat leaf-list experiment (CPU): how many leaf boxes does a real path segment's ray enter?  python tests/lab/flat_lab.py [scene] [max_path]"""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)
import numpy as np
import run_lab
run_lab.build()
import oracle, master_amd as ma
oracle.ORACLE_LIB = run_lab.SO; oracle.build = lambda: run_lab.SO
L = oracle.lib()
L.lab_flat_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]; L.lab_flat_size.restype = C.c_uint64
name = sys.argv[1] if len(sys.argv) > 1 else "CornellBoxDiffuse"
mp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
s = ma.Scene.load(os.path.join(ROOT, "scenes", name + ".miscene"))
o = oracle.Oracle(s, max_path=mp)
buf = np.zeros(1 << 22, np.uint32)
k = L.lab_flat_begin(o._h, buf.ctypes.data_as(C.c_void_p), buf.size)
rng = np.random.default_rng(1); n = 20000; W = H = 128
xy = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.uint32); si = rng.integers(0, 64, n).astype(np.uint64)
o.trace_paths(W, H, xy, si, seed=3)
m = int(L.lab_flat_size()); ev = buf[:m]
cl = ev[(ev & 1) == 1]; sh = ev[(ev & 1) == 0]
ent = (cl >> 4) & 255; ent2 = (cl >> 20) & 255
print("%s: %d leaf links; %d closest rays, %d shadow rays" % (name, k, len(cl), len(sh)))
def emax(a, reps=2000):
    idx = rng.integers(0, len(a), (reps, 64)); return a[idx].max(1).mean()
print("closest: boxes entered mean %.2f  E[max of 64] %.2f  hist %s" % (ent.mean(), emax(ent), np.bincount(ent)[:12]))
print("closest, after testing the nearest box first: further boxes entered with tmax = hit: mean %.2f E[max64] %.2f hist %s" % (ent2.mean(), emax(ent2), np.bincount(ent2)[:12]))
se = (sh >> 4) & 255; st = (sh >> 12) & 255
print("shadow: boxes entered mean %.2f E[max64] %.2f; leaves tested until first hit mean %.2f E[max64] %.2f; occluded %.2f" % (se.mean(), emax(se), st.mean(), emax(st), ((sh >> 28) & 1).mean()))
