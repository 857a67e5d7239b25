#!/usr/bin/env python3
"""Generates tests/golden/*.json.  The reference executable cannot be built here (DESIGN.md,
Oracle), so these vectors come from (a) the reference's own test constants — copied as numbers
with their source line — and (b) the CPU oracle, frozen so later edits of the oracle or of the
device code are caught (regression pins, not reference outputs; each file says which)."""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import master_amd as ma  # noqa: E402
import oracle  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)


def main():
    # (a) reference test constants
    json.dump({
        "source": "reference test constants (numbers only)",
        "fovy_aspect_4_3_fovx_pi_2": {"value": 1.2870022, "from": "unit_tests/Cameras.test.cpp:25"},
        "fovy_aspect_1_fovx_pi_2": {"value": 1.5707963267948966, "from": "unit_tests/Cameras.test.cpp:24"},
        "round_trip": {"resolution": [800.0, 600.0], "fov_y": 1.5707963267948966, "position": [123.4, 345.0], "from": "Cameras.cpp:164-173"},
        "centre_ray_800x600": {"direction": [0.0, 0.0, -1.0], "from": "unit_tests/Cameras.test.cpp:40"},
        "corner_ray_aspect1": {"direction": [-0.577, -0.577, -0.577], "tol": 1e-3, "from": "unit_tests/Cameras.test.cpp:47"},
        "test_scene_mean_radiance": {"value": 1.0, "scenes": ["TestCase0", "TestCase2", "TestCase5", "TestCase25", "TestCaseFurnace"],
                                     "from": "models/TestCase*.blend are normalised by their author so the 512x512 image averages 1 (lamp energies such as 53.9002 are tuned); unit_test.py:77-83 ranks renders by distance of the average from a constant"},
    }, open(os.path.join(G, "reference_constants.json"), "w"), indent=1)

    # (b) oracle regression pins
    rng = {"source": "oracle regression pin (PCG32 / splitmix64 stream definition of this build)", "cases": []}
    for seed, pixel, sample in [(0, 0, 0), (0x5EED, 131071, 1023), (2 ** 63 + 12345, 0xFFFFFFFF, 2 ** 40 + 7)]:
        v = oracle.rng_floats(seed, pixel, sample, 8)
        rng["cases"].append({"seed": seed, "pixel": pixel, "sample": sample, "floats_hex": [float(x).hex() for x in v]})
    json.dump(rng, open(os.path.join(G, "rng_kat.json"), "w"), indent=1)

    s = ma.Scene.load(os.path.join(ROOT, "scenes", "CornellBoxDiffuse.miscene"))
    pins = {"source": "oracle regression pin: CornellBoxDiffuse 32x32, per-path radiance of samples 0..3, seed 7", "cases": []}
    for max_path in (1, 2, 4, 8):
        o = oracle.Oracle(s, max_path=max_path)
        xy = np.stack(np.meshgrid(np.arange(32), np.arange(32)), -1).reshape(-1, 2).astype(np.uint32)
        xy = np.tile(xy, (4, 1)); si = np.repeat(np.arange(4, dtype=np.uint64), 1024)
        rad, cnt = o.trace_paths(32, 32, xy, si, seed=7)
        pins["cases"].append({"max_path": max_path, "sum_radiance_hex": [float(x).hex() for x in rad.astype(np.float64).sum(0)],
                              "xor_bits": int(np.bitwise_xor.reduce(rad.view(np.uint32).ravel())), "basic": int(cnt[:, 0].sum()), "shadow": int(cnt[:, 1].sum())})
    json.dump(pins, open(os.path.join(G, "cornell_paths_pin.json"), "w"), indent=1)
    nodes, order, morton = oracle.Oracle(s).bvh()
    json.dump({"source": "oracle regression pin: LBVH of CornellBoxDiffuse", "sorted_tri": order.tolist(), "morton": morton.tolist(),
               "links": [[int(n["link0"]), int(n["link1"])] for n in nodes]}, open(os.path.join(G, "cornell_lbvh_pin.json"), "w"), indent=1)
    print("golden written to", G)


if __name__ == "__main__":
    main()
