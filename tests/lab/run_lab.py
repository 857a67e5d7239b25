#!/usr/bin/env python3
"""Tree-quality lab (CPU): node visits / triangle tests per ray over real path segments for BVH variants.
    python tests/lab/run_lab.py CornellBoxDiffuse MetalRings atrium:60000
"""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

SO = os.path.join(HERE, "_lab.so")


def build():
    src = os.path.join(HERE, "bvh_lab.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "oracle", "pt_oracle.c"))):
        subprocess.check_call(["cc", "-O2", "-std=gnu11", "-mfma", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-o", SO, src, "-lm"])


def main():
    build()
    import oracle
    import master_amd as ma
    from master_amd import scenegen
    oracle.ORACLE_LIB = SO; oracle.build = lambda: SO  # the binding resolves every orc_* symbol from the lab build
    L = oracle.lib()
    L.lab_rotate.argtypes = [C.c_void_p, C.c_int]; L.lab_build_sah.argtypes = [C.c_void_p]; L.lab_build_ploc_refs.argtypes = [C.c_void_p]; L.lab_sah.argtypes = [C.c_void_p]
    L.lab_sah.restype = C.c_double
    specs = sys.argv[1:] or ["CornellBoxDiffuse", "CornellBoxSpecular", "MirrorBalls", "MetalRings", "LivingRoomLit", "atrium:60000"]
    for spec in specs:
        p = os.path.join(ROOT, "scenes", spec + ".miscene")
        s = ma.Scene.load(p) if os.path.exists(p) else scenegen.load(spec)
        rng = np.random.default_rng(1); n = 6000
        W, H = 320, 180
        xy = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.uint32); si = rng.integers(0, 64, n).astype(np.uint64)
        for variant in os.environ.get("LAB_VARIANTS", "lbvh,ploc,ploc+rot,lbvh+rot,sah,sah+rot").split(","):
            os.environ["MI_PT_BVH"] = "lbvh" if variant.startswith("lbvh") else "ploc"
            o = oracle.Oracle(s)
            if variant.startswith("sah") or variant.startswith("plocr"):
                tail = variant[3:] if variant.startswith("sah") else variant[5:]
                L.lab_set_split(int(tail[0]) if tail and tail[0].isdigit() else 0)
                L.lab_set_split_big(int(variant.split("big")[1]) if "big" in variant else 0)
                (L.lab_build_sah if variant.startswith("sah") else L.lab_build_ploc_refs)(o._h)
            rot = L.lab_rotate(o._h, 8) if variant.endswith("+rot") else 0
            cnt = (C.c_uint64 * 4)()
            L.lab_counters(cnt, 1)
            _, rc = o.trace_paths(W, H, xy, si, seed=3)
            L.lab_counters(cnt, 1)
            basic, shadow = rc[:, 0].sum(), max(1, rc[:, 1].sum())
            info = o.bvh_info()
            print("%-20s %-9s depth %3d SAH %7.2f  N %6.2f T %5.2f | N' %6.2f T' %5.2f (per shadow ray)  cost %7.1f  rot %d" % (
                spec, variant, info.max_depth, L.lab_sah(o._h), cnt[0] / basic, cnt[1] / basic, cnt[2] / shadow, cnt[3] / shadow,
                (cnt[0] + 1.5 * cnt[1] + cnt[2] + 1.5 * cnt[3]) / basic, rot))


if __name__ == "__main__":
    main()
