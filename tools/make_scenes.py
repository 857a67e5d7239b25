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
          "TestCase3", "TestCase5", "TestCase6", "TestCase7", "TestCase25", "SingleAreaLight", "DoubleLight", "MirrorAndAreaLight",
          "MirrorBalls", "MetalRings",  # MetalRings: 30 558 triangles, Phong, two area lights — largest lit model present
          # every other lit model of the reference below 1 MB as a fixture (parity corpus: sun lights, glass, Phong, mirrors)
          "IndirectCubeEye", "IndirectCubeH25", "IndirectCubeHalf", "IndirectCubeIOR1", "IndirectCubeIOR2", "IndirectCubeLens1",
          "IndirectCubeNone", "IndirectSphereIOR2", "LightNoOcclusionTest", "LightOverBox", "LightPathNone", "LightPathShaded",
          "RandomNumberGeneratorTest", "SimpleCubeIOR1", "SimpleSphereEmpty", "SimpleSphereIOR1", "SimpleSphereIOR2",
          "TestCase4", "TestCase8", "TestCase9", "TestCase10", "TestCase11", "TestCase12", "TestCase13", "TestCase14", "TestCase15",
          "TestCase16", "TestCase17", "TestCase18", "TestCase23", "TestCase24", "TestCase27", "TestCase29", "TestCase30", "TestCase31",
          "TestCase32", "TestCase33", "TestCase35", "TestCase37", "TestCase38", "TestCase39", "TestCase41", "TransparentBoxAndSphere"]

# LivingRoom.blend (43 944 triangles, the largest model present) has no lamps; the reference cannot light it with PT
# either.  The fixture keeps its geometry, materials and camera and adds ONE area light under the ceiling lamp.
LIT = {"LivingRoom": dict(position=(0.383, -2.893, 2.0), direction=(0, 0, -1), up=(0, 1, 0), size=(0.8, 0.8), exitance=(60.0, 54.0, 45.0))}


def with_area_light(s, position, direction, up, size, exitance):
    """Scene `s` + one area light, appended the way loader.cpp:434-456 appends light quads after the meshes."""
    import numpy as np
    from master_amd import scenegen
    b = scenegen.Builder()
    b.pos, b.tan = list(s.positions), list(s.tangents)
    b.idx = list(s.indices.reshape(-1))
    b.off, b.mesh_mat = list(s.mesh_tri_offset), list(s.mesh_material_id)
    b.materials, b.lights, b.cameras = list(s.materials), list(s.lights), list(s.cameras)
    b.add_light(position, direction, up, size, exitance)
    return b.build()


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/models"
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes")
    os.makedirs(dst, exist_ok=True)
    for name in SCENES + list(LIT):
        p = os.path.join(src, name + ".blend")
        if not os.path.exists(p):
            print("missing", p)
            continue
        s = ma.Scene.load_blend(p)
        if name in LIT:
            s = with_area_light(s, **LIT[name])
            name += "Lit"
        out = os.path.join(dst, name + ".miscene")
        s.save(out)
        kinds = sorted(set(m.type for m in s.materials))
        print("%-22s tris %6d lights %d cameras %d bsdf kinds %s -> %d bytes" % (name, s.n_triangles, len(s.lights), len(s.cameras), kinds, os.path.getsize(out)))


if __name__ == "__main__":
    main()
