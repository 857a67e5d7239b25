"""CPU tests of the BPT restatement (oracle/bpt_oracle.inc) — the checker for the next row of SURVEY.md 8(f).
The reference has no executable BPT tests; its unit_test.py renders models/TestCase*.blend with BPT and compares the image
average with a constant: those models are normalised by their author to an average of 1.  That, and agreement with the PT
restatement (same expectation, different estimator), pin this oracle."""
import os

import numpy as np
import pytest

import master_amd as ma
import oracle
from conftest import REFERENCE, load_scene


def _mean(img):
    return float((img[..., :3] / np.maximum(img[..., 3:], 1)).mean())


@pytest.mark.parametrize("name,beta", [("TestCase0", 2.0), ("TestCase2", 1.0), ("TestCase9", 2.0), ("TestCase25", 0.0), ("TestCaseFurnace", 2.0),
                                       ("TestCase10", 2.0), ("TestCase13", 1.0), ("TestCase18", 2.0), ("TestCase29", 2.0), ("TestCase33", 1.5)])
def test_normalised_models_average_one(name, beta):
    """Area lights (TestCase0/2/9/25, furnace) and sun lights (TestCase10/13/18/29/33: PT cannot light these at all) — the
    bidirectional estimator with every beta variant of Beta.hpp averages 1 over the image."""
    s = load_scene(name)
    img = oracle.Oracle(s, beta=beta).bpt_render_rgbn(48, 48, spp=48, seed=3, threads=8)
    assert np.all(img[..., 3] == 48) and np.isfinite(img).all()
    assert abs(_mean(img) - 1.0) < 0.03, _mean(img)


@pytest.mark.parametrize("parent,copies", [("TestCase31", ["TestCase35"]), ("TestCase33", ["TestCase37", "TestCase38", "TestCase39", "TestCase41"]),
                                            ("TestCase30", ["TestCase32"])])
def test_models_that_miss_the_constant_are_untuned_copies_of_normalised_ones(parent, copies):
    """VERDICT r02 #3.  The author tuned each model's lamp energy until its image average reached unit_test.py's constant (energies like 53.9002,
    775.314 ...).  TestCase35 / 37 / 38 / 39 / 41 (and 32) average 0.29 ... 1.33 through this build — but each carries, BIT FOR BIT, the energy of an
    earlier model that does average 1.00 (31 -> 34, 35; 33 -> 36 ... 43; 30 -> 32; tools/testcase_energies.py over the reference's .blend files,
    profiles/r03/testcase_lamp_energies.txt): copies with edited geometry or camera that were saved without re-normalising.  Nothing in the reader
    (loader.cpp:293-456) or the sun-light terms (BSDF.cpp:164-193, BPT.cpp:192-225) is left to explain — the parent, read and rendered by the same
    code, is on the constant."""
    p = load_scene(parent)
    img = oracle.Oracle(p, beta=2.0).bpt_render_rgbn(64, 64, spp=256, seed=3, threads=8)  # TestCase30's sun light is a high-variance case: 0.02 of spread at 128 spp
    assert abs(_mean(img) - 1.0) < 0.04, _mean(img)
    for c in copies:
        q = load_scene(c)
        assert len(q.lights) == len(p.lights) == 1
        assert np.array_equal(np.array(list(q.lights[0].exitance), np.float32).view(np.uint32), np.array(list(p.lights[0].exitance), np.float32).view(np.uint32))
        assert list(q.lights[0].size) == list(p.lights[0].size) and q.lights[0].diffuse == p.lights[0].diffuse


@pytest.mark.parametrize("name", ["CornellBoxDiffuse", "CornellBoxSpecular", "MirrorAndAreaLight"])
def test_bpt_and_pt_agree(name):
    s = load_scene(name)
    o = oracle.Oracle(s, beta=2.0)
    b = o.bpt_render_rgbn(48, 48, spp=96, seed=1, threads=8); p = o.render_rgbn(48, 48, spp=96, seed=2, threads=8)
    assert abs(_mean(b) - _mean(p)) / _mean(p) < 0.03


def test_paths_are_deterministic_and_splats_stay_in_the_image(cornell):
    o = oracle.Oracle(cornell, beta=2.0)
    rng = np.random.default_rng(5); n = 4000
    xy = np.stack([rng.integers(0, 64, n), rng.integers(0, 48, n)], 1).astype(np.uint32); si = rng.integers(0, 32, n).astype(np.uint64)
    r1, s1, c1 = o.bpt_trace_paths(64, 48, xy, si, seed=7); r2, s2, c2 = o.bpt_trace_paths(64, 48, xy, si, seed=7)
    assert np.array_equal(r1, r2) and np.array_equal(s1, s2) and np.array_equal(c1, c2)
    assert np.isfinite(r1).all() and np.isfinite(s1).all()
    # the light sub-path is connected to the camera at most once per vertex: splats <= shadow rays; paths killed by the first roulette cast nothing
    assert (c1[:, 2] <= c1[:, 1]).all()
    dead = c1[:, 0] == 0
    assert 0.05 < dead.mean() < 0.15 and not r1[dead].any() and not c1[dead].any()
    img = o.bpt_render_rgbn(64, 48, spp=4, seed=7, threads=4)
    assert np.all(img[..., 3] == 4)


def test_sun_lights_are_invisible_to_pt_but_not_to_bpt():
    s = load_scene("TestCase12")
    o = oracle.Oracle(s, beta=2.0)
    assert _mean(o.render_rgbn(32, 32, spp=32, seed=1, threads=8)) < 0.1
    assert abs(_mean(o.bpt_render_rgbn(32, 32, spp=64, seed=1, threads=8)) - 1.0) < 0.08


# ---- the radiometric scale of the importer ----
_REF_MODELS = os.path.join(REFERENCE, "models")
needs_reference = pytest.mark.skipif(not os.path.isdir(_REF_MODELS), reason="reference tree not present (GPU box)")


@needs_reference
@pytest.mark.parametrize("name", ["TestCase0", "TestCase9", "TestCaseFurnace", "TestCase10"])
def test_lamp_energy_scale_of_one_hundredth_reproduces_the_protocol_constant(name):
    """unit_test.py:77-83 steers every TestCase*.blend toward an image average of `expected = [0.01] * 3` (exr_average, exr.cpp:315-340).  Through this
    build's reader with stock-assimp lamp units (exitance = rgb * energy, loader.cpp:434-456; mi_blend_options.lamp_energy_scale = 1, the default) the
    normalised models average 1.000 — a clean factor of 100.  With lamp_energy_scale = 0.01 the same models average the reference's own constant,
    0.0100 +- 0.0002, area lights (TestCase0 / 9, furnace) and sun lights (TestCase10) alike.  Which of the two the assimp fork implements cannot be
    decided without running the reference (the absolute scale of the importer stays unpinned); the default stays at stock assimp's 1.0 and the
    other value is one option away (include/mi_pt.h)."""
    s = ma.Scene.load_blend(os.path.join(_REF_MODELS, name + ".blend"), lamp_energy_scale=0.01)
    img = oracle.Oracle(s, beta=2.0).bpt_render_rgbn(64, 64, spp=64, seed=3, threads=8)
    assert np.all(img[..., 3] == 64) and np.isfinite(img).all()
    assert abs(_mean(img) - 0.01) < 0.0002, _mean(img)


@needs_reference
def test_lamp_energy_scale_is_a_linear_factor_of_the_image():
    """The option multiplies every lamp's exitance and nothing else: light selection (proportional to power), MIS weights (ratios of densities) and
    every path are unchanged, so the PT image scales by the factor up to the rounding of the products."""
    path = os.path.join(_REF_MODELS, "CornellBoxDiffuse.blend")
    one, hundredth = ma.Scene.load_blend(path), ma.Scene.load_blend(path, lamp_energy_scale=0.01)
    assert len(one.lights) == len(hundredth.lights) == 1
    np.testing.assert_allclose(np.array(list(hundredth.lights[0].exitance)), 0.01 * np.array(list(one.lights[0].exitance)), rtol=2e-7)
    a = oracle.Oracle(one, max_path=6).render_rgbn(40, 32, spp=6, seed=9, threads=8)
    b = oracle.Oracle(hundredth, max_path=6).render_rgbn(40, 32, spp=6, seed=9, threads=8)
    assert np.array_equal(a[..., 3], b[..., 3])
    np.testing.assert_allclose(b[..., :3], 0.01 * a[..., :3], rtol=2e-6, atol=1e-12)
