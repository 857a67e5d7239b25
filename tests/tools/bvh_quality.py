#!/usr/bin/env python3
"""SAH cost of the oracle's BVH (bit-identical to the device's) for both builders.

    python tools/bvh_quality.py CornellBoxDiffuse MetalRings atrium:60000 ...
cost = sum over child boxes of area / root area; internal children weigh 1 (a node visit), leaf children 1 (a triangle test).
"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def measure(spec):
    import numpy as np
    import oracle
    import master_amd as ma
    from master_amd import scenegen
    p = os.path.join(ROOT, "scenes", spec + ".miscene")
    s = ma.Scene.load(p) if os.path.exists(p) else scenegen.load(spec)
    o = oracle.Oracle(s)
    info = o.bvh_info()
    nodes, _, _ = o.bvh()

    def area(lo, hi):
        d = (hi - lo).astype(np.float64)
        return d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]
    a0, a1 = area(nodes["lo0"], nodes["hi0"]), area(nodes["lo1"], nodes["hi1"])
    root = area(np.minimum(nodes["lo0"][:1], nodes["lo1"][:1]), np.maximum(nodes["hi0"][:1], nodes["hi1"][:1]))[0]
    inner = (a0 * (nodes["link0"] >= 0)).sum() + (a1 * (nodes["link1"] >= 0)).sum()
    leaf = (a0 * (nodes["link0"] < 0)).sum() + (a1 * (nodes["link1"] < 0)).sum()
    print("%-22s builder %s tris %8d depth %3d rounds %3d  SAH nodes %8.2f leaves %7.2f" % (
        spec, "ploc" if info.builder else "lbvh", info.n_triangles, info.max_depth, info.build_rounds, 1.0 + inner / root, leaf / root))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        measure(sys.argv[2])
    else:
        for spec in sys.argv[1:] or ["CornellBoxDiffuse", "CornellBoxSpecular", "MirrorBalls", "MetalRings", "LivingRoomLit"]:
            for b in ("lbvh", "ploc"):
                subprocess.run([sys.executable, __file__, "--one", spec], env=dict(os.environ, MI_PT_BVH=b), check=True)
