#!/usr/bin/env python3
"""CLI wrapper: prints triangle counts of the procedural stand-in scenes (master_amd/scenegen.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from master_amd import scenegen  # noqa: E402

load, SCENES = scenegen.load, scenegen.SCENES

if __name__ == "__main__":
    for spec in sys.argv[1:] or ["atrium", "clutter"]:
        s = load(spec)
        print(spec, "triangles", s.n_triangles, "materials", len(s.materials), "lights", len(s.lights))
