#!/usr/bin/env python3
"""Converts the reference's .blend models (data files, /root/reference/models) into flat
.miscene fixtures under scenes/ with the build's own .blend reader, so tests, smoke() and
bench.py can run where the reference tree does not exist (the GPU box).

    python tools/make_scenes.py [/root/reference/models]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import master_amd as ma  # noqa: E402

SCENES = ["CornellBoxDiffuse", "CornellBoxPhong", "CornellBoxSpecular", "TestCaseFurnace", "TestCase0", "TestCase1", "TestCase2",
          "TestCase3", "TestCase5", "TestCase6", "TestCase7", "TestCase25", "SingleAreaLight", "DoubleLight", "MirrorAndAreaLight"]


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/models"
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes")
    os.makedirs(dst, exist_ok=True)
    for name in SCENES:
        p = os.path.join(src, name + ".blend")
        if not os.path.exists(p):
            print("missing", p)
            continue
        s = ma.Scene.load_blend(p)
        out = os.path.join(dst, name + ".miscene")
        s.save(out)
        kinds = sorted(set(m.type for m in s.materials))
        print("%-22s tris %6d lights %d cameras %d bsdf kinds %s -> %d bytes" % (name, s.n_triangles, len(s.lights), len(s.cameras), kinds, os.path.getsize(out)))


if __name__ == "__main__":
    main()
