"""GPU parity of the BPT kernels against the BPT oracle, through the C ABI (mi_bpt_*).
Bit-exact per path — Phong lobes and VariableBeta included, pow() being the build's own mi_powf on both sides: eye-image
radiance, the sum of the light-image splats, closest-hit / shadow ray counts and the number of splats."""
import os

import numpy as np
import pytest

import master_amd as ma
import oracle
from conftest import load_scene

pytestmark = pytest.mark.gpu


def _paths(w, h, n, seed):
    rng = np.random.default_rng(seed)
    return np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], 1).astype(np.uint32), rng.integers(0, 32, n).astype(np.uint64)


def _bits_equal(a, b):
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all(1)


def _corpus():
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes")
    return sorted(f[:-8] for f in os.listdir(d) if f.endswith(".miscene") and os.path.getsize(os.path.join(d, f)) < 1000000)


@pytest.mark.parametrize("name", _corpus())
def test_bpt_reference_corpus_parity(name):
    s = load_scene(name)
    pt, orc = ma.PathTracing(s, beta=2.0), oracle.Oracle(s, beta=2.0)
    xy, si = _paths(64, 48, 5000, 5)
    gr, gs, gc = pt.bpt_trace_paths(64, 48, xy, si, seed=7); orr, os_, oc = orc.bpt_trace_paths(64, 48, xy, si, seed=7)
    assert np.array_equal(gc, oc)
    assert _bits_equal(gr, orr).all() and _bits_equal(gs, os_).all()


@pytest.mark.parametrize("beta", [0.0, 1.0, 2.0, 1.5])
def test_bpt_beta_variants(cornell, beta):
    pt, orc = ma.PathTracing(cornell, beta=beta, roulette=0.8), oracle.Oracle(cornell, beta=beta, roulette=0.8)
    xy, si = _paths(48, 48, 6000, 9)
    gr, gs, gc = pt.bpt_trace_paths(48, 48, xy, si, seed=3); orr, os_, oc = orc.bpt_trace_paths(48, 48, xy, si, seed=3)
    assert np.array_equal(gc, oc)
    assert _bits_equal(gr, orr).all() and _bits_equal(gs, os_).all()  # VariableBeta (1.5): mi_powf on both sides


@pytest.mark.parametrize("name,w,h,window", [("CornellBoxDiffuse", 64, 48, None), ("TestCase10", 37, 23, None), ("CornellBoxSpecular", 40, 40, None),
                                             ("CornellBoxDiffuse", 64, 48, (5, 7, 21, 30)), ("TestCase0", 130, 9, (120, 0, 10, 9))])
def test_bpt_render_equals_oracle(name, w, h, window):
    """Frames: eye image + light image, one finite filter per pixel and frame (Technique.cpp:194-244); ragged sizes and view
    windows (only the window's pixels trace paths and are committed; their splats may land anywhere)."""
    s = load_scene(name)
    pt, orc = ma.PathTracing(s, beta=2.0), oracle.Oracle(s, beta=2.0)
    img = pt.bpt_render_rgbn(w, h, spp=12, seed=5, sample_offset=2, window=window); ref = orc.bpt_render_rgbn(w, h, spp=12, seed=5, sample_offset=2, threads=8, window=window)
    if window:
        x0, y0, ww, hh = window
        mask = np.ones((h, w), bool); mask[y0:y0 + hh, x0:x0 + ww] = False
        assert not img[mask].any()
    st, so = pt.last_stats, orc.last_stats
    assert (st.num_paths, st.num_basic_rays, st.num_shadow_rays, st.numeric_errors) == (so.num_paths, so.num_basic_rays, so.num_shadow_rays, so.numeric_errors)
    assert np.array_equal(img[..., 3], ref[..., 3])
    np.testing.assert_allclose(img, ref, rtol=2e-6, atol=0)  # FP64 splat order is free; everything else is exact
    again = pt.bpt_render_rgbn(w, h, spp=12, seed=5, sample_offset=2, window=window)
    np.testing.assert_allclose(img, again, rtol=2e-6, atol=0)


NORMALISED = ["TestCase0", "TestCase1", "TestCase2", "TestCase3", "TestCase4", "TestCase5", "TestCase6", "TestCase7", "TestCase8", "TestCase9", "TestCase10",
              "TestCase12", "TestCase13", "TestCase14", "TestCase15", "TestCase16", "TestCase17", "TestCase18", "TestCase23", "TestCase24", "TestCase25",
              "TestCase29", "TestCase30", "TestCase31", "TestCase33", "TestCaseFurnace"]


@pytest.mark.parametrize("name", NORMALISED)
def test_bpt_normalised_models_average_one_on_device(name):
    """The reference's own test protocol (unit_test.py: BPT on models/TestCase*.blend, image average against a constant): these
    models are normalised to an average of 1 — area and sun lamps, diffuse and Phong (profiles/r01/bpt_testcase_averages.txt)."""
    s = load_scene(name)
    img = ma.PathTracing(s, beta=2.0).bpt_render_rgbn(128, 128, spp=256, seed=1)
    m = float((img[..., :3] / np.maximum(img[..., 3:], 1)).mean())
    assert abs(m - 1.0) < 0.015, m


@pytest.mark.parametrize("name,expected", [("TestCase11", 0.5)])  # TestCase32 converges to 0.2502 at 256^2 x 512 but is heavy-tailed: not a unit test
def test_bpt_half_and_quarter_cases(name, expected):
    img = ma.PathTracing(load_scene(name), beta=2.0).bpt_render_rgbn(128, 128, spp=512, seed=1)
    assert abs(float((img[..., :3] / np.maximum(img[..., 3:], 1)).mean()) - expected) < 0.03 * expected + 0.005  # a high-variance caustic case


def test_bpt_and_pt_converge_to_the_same_image(cornell):
    pt = ma.PathTracing(cornell, beta=2.0, max_path=ma.PTRDIFF_MAX)
    b = pt.bpt_render_rgbn(64, 64, spp=512, seed=1); p = pt.render_rgbn(64, 64, spp=512, seed=2)
    br, pr = b[..., :3] / b[..., 3:], p[..., :3] / p[..., 3:]
    assert abs(br.mean() - pr.mean()) / pr.mean() < 0.01
    assert np.sqrt(np.mean((br - pr) ** 2)) < 0.05


@pytest.mark.parametrize("name", ["CornellBoxDiffuse", "MirrorAndAreaLight", "TestCaseFurnace", "TestCase10"])
def test_bpt_one_kernel_form_equals_staged_form(monkeypatch, name):
    """MI_BPT_STAGED=0 runs the whole path in one lane; the staged default cuts it into trace / items / gather.  Same terms,
    same order of every float sum: identical bits."""
    s = load_scene(name)
    pt = ma.PathTracing(s, beta=2.0)
    xy, si = _paths(64, 48, 8000, 11)
    a = pt.bpt_trace_paths(64, 48, xy, si, seed=5)
    monkeypatch.setenv("MI_BPT_STAGED", "0")
    b = pt.bpt_trace_paths(64, 48, xy, si, seed=5)
    assert np.array_equal(a[2], b[2]) and _bits_equal(a[0], b[0]).all() and _bits_equal(a[1], b[1]).all()
    img_b = pt.bpt_render_rgbn(40, 30, spp=6, seed=2)
    monkeypatch.delenv("MI_BPT_STAGED")
    img_a = pt.bpt_render_rgbn(40, 30, spp=6, seed=2)
    assert np.array_equal(img_a[..., 3], img_b[..., 3])
    np.testing.assert_allclose(img_a, img_b, rtol=2e-6, atol=0)


@pytest.mark.parametrize("name", ["CornellBoxDiffuse", "CornellBoxPhong", "TestCaseFurnace", "TestCase10", "TestCase12", "MirrorAndAreaLight", "IndirectCubeNone"])
def test_bpt_flat_leaf_list_is_bit_identical_per_path(monkeypatch, name):
    """r03: the staged BPT kernels of LDS-resident scenes walk the flat leaf list too (traverse_flat: closest hits under a geometry mask —
    Scene::intersectMesh — and any-hit shadow rays).  Forced on and off: eye radiance, splat sums, ray and splat counts identical to the tree walk and
    to the oracle; sun lights, mirrors and tables of 3..29 leaves included."""
    s = load_scene(name)
    pt = ma.PathTracing(s, beta=2.0)
    xy, si = _paths(64, 48, 8000, 11)
    monkeypatch.setenv("MI_BPT_FLAT", "0")
    a = pt.bpt_trace_paths(64, 48, xy, si, seed=5)
    img_a = pt.bpt_render_rgbn(40, 30, spp=6, seed=2)
    monkeypatch.setenv("MI_BPT_FLAT", "1")
    b = pt.bpt_trace_paths(64, 48, xy, si, seed=5)
    img_b = pt.bpt_render_rgbn(40, 30, spp=6, seed=2)
    assert np.array_equal(a[2], b[2]) and _bits_equal(a[0], b[0]).all() and _bits_equal(a[1], b[1]).all()
    assert np.array_equal(img_a[..., 3], img_b[..., 3])
    np.testing.assert_allclose(img_a, img_b, rtol=2e-6, atol=0)
    o = oracle.Oracle(s, beta=2.0).bpt_trace_paths(64, 48, xy, si, seed=5)
    assert np.array_equal(o[2], b[2]) and _bits_equal(o[0], b[0]).all() and _bits_equal(o[1], b[1]).all()


@pytest.mark.parametrize("name,wide", [("CornellBoxDiffuse", 0), ("CornellBoxSpecular", 0), ("CornellBoxSpecular", 1), ("LivingRoomLit", 0), ("LivingRoomLit", 1),
                                       ("MirrorAndAreaLight", 0), ("TestCaseFurnace", 0), ("TestCase10", 1), ("TestCase29", 0)])
def test_bpt_visibility_stage_is_bit_identical_per_path(monkeypatch, name, wide):
    """r02: the connections' shadow rays as a list walked by persistent waves that refill idle lanes (bpt_rays / bpt_visibility), against
    every item walking its own ray inside bpt_items.  Occlusion is a boolean of the ray: same values, same ray counts, same splat sums.
    The stage is on by default only for large launches over scenes read from HBM; MI_BPT_DYN_VIS forces it (LDS-resident scenes, binary and
    wide quantised nodes, sun lights = items without a shadow ray, splats outside the image)."""
    s = load_scene(name)
    pt = ma.PathTracing(s, beta=2.0)
    xy, si = _paths(64, 48, 12000, 13)
    monkeypatch.setenv("MI_BPT_DYN_VIS", "0")
    a = pt.bpt_trace_paths(64, 48, xy, si, seed=5)
    img_a = pt.bpt_render_rgbn(40, 30, spp=6, seed=2)
    monkeypatch.setenv("MI_BPT_DYN_VIS", "1"); monkeypatch.setenv("MI_BPT_VIS_WIDE", str(wide))
    b = pt.bpt_trace_paths(64, 48, xy, si, seed=5)
    img_b = pt.bpt_render_rgbn(40, 30, spp=6, seed=2)
    assert np.array_equal(a[2], b[2]) and _bits_equal(a[0], b[0]).all() and _bits_equal(a[1], b[1]).all()
    assert np.array_equal(img_a[..., 3], img_b[..., 3])
    np.testing.assert_allclose(img_a, img_b, rtol=2e-6, atol=0)  # FP64 splat order is free
    assert pt.last_stats.num_shadow_rays > 0 or name == "TestCase10"


@pytest.mark.parametrize("name,wide", [("LivingRoomLit", 1), ("CornellBoxSpecular", 1), ("MetalRings", 0), ("TestCase10", 1)])
@pytest.mark.parametrize("rounds", ["0", "3", "60"])
def test_bpt_tracing_stage_as_uniform_steps_is_bit_identical_per_path(monkeypatch, name, wide, rounds):
    """r04: the tracing stage as uniform steps — a path is a coroutine suspended at every closest-hit ray; bpt_step resumes the paths that
    have a hit waiting and runs them to their next ray, persistent waves with lane refill (bpt_closest) walk the rays of a round, and after
    MI_BPT_STEP_ROUNDS rounds the tail kernel lets the paths still in flight run to their ends.  Same draws in the same order, same records: eye
    radiance, splat sums and ray counts per path equal the per-lane form's (and with it the oracle's), whatever the number of rounds (none: every path
    ends in the tail; 60: nearly every path ends in a round).  Measured slower than the per-lane form so far (profiles/r04/ab_bpt_steps.txt): opt-in."""
    s = load_scene(name)
    pt = ma.PathTracing(s, beta=2.0)
    xy, si = _paths(64, 48, 12000, 17)
    monkeypatch.setenv("MI_PT_WIDE_NODES", str(wide))
    monkeypatch.setenv("MI_BPT_STEPS", "0")
    a = pt.bpt_trace_paths(64, 48, xy, si, seed=6)
    img_a = pt.bpt_render_rgbn(40, 30, spp=5, seed=2)
    monkeypatch.setenv("MI_BPT_STEPS", "1"); monkeypatch.setenv("MI_BPT_STEP_ROUNDS", rounds)
    b = pt.bpt_trace_paths(64, 48, xy, si, seed=6)
    img_b = pt.bpt_render_rgbn(40, 30, spp=5, seed=2)
    assert np.array_equal(a[2], b[2]) and _bits_equal(a[0], b[0]).all() and _bits_equal(a[1], b[1]).all()
    assert np.array_equal(img_a[..., 3], img_b[..., 3])
    np.testing.assert_allclose(img_a, img_b, rtol=2e-6, atol=0)  # FP64 splat order is free
    # the same coroutine as PASSES with growing ray budgets (MI_BPT_STEPS=2): every path walks its own rays but is suspended after `budget` rays of a pass
    monkeypatch.setenv("MI_BPT_STEPS", "2"); monkeypatch.setenv("MI_BPT_PASS_CAPS", {"0": "3", "3": "2,5,11", "60": "40"}[rounds])
    c = pt.bpt_trace_paths(64, 48, xy, si, seed=6)
    assert np.array_equal(a[2], c[2]) and _bits_equal(a[0], c[0]).all() and _bits_equal(a[1], c[1]).all()
    if rounds == "3" and name == "CornellBoxSpecular":  # and against the oracle directly
        o = oracle.Oracle(s, beta=2.0).bpt_trace_paths(64, 48, xy[:3000], si[:3000], seed=6)
        assert np.array_equal(np.asarray(b[2])[:3000], np.asarray(o[2])) and _bits_equal(b[0][:3000], o[0]).all() and _bits_equal(b[1][:3000], o[1]).all()


@pytest.mark.parametrize("name,wide", [("LivingRoomLit", 1), ("CornellBoxSpecular", 1), ("MetalRings", 0), ("TestCase10", 1)])
def test_bpt_tracing_stage_with_path_regeneration_is_bit_identical_per_path(monkeypatch, name, wide):
    """r04: the tracing stage as two kernels of resident waves (bpt_trace_light, bpt_trace_eye): a trip extends every live sub-path by one vertex and a lane
    whose sub-path has ended takes the launch's next path from a cursor; between the kernels a path's generator state and light sub-path length wait in its
    info record.  Whatever lane walks a path, its draws, records and counts are the per-lane form's (MI_BPT_PERSIST=0) and the oracle's.  Default for the
    models walked at six waves per SIMD (LivingRoomLit 205 -> 188 ms per 64 frames), forced here on every model read from HBM."""
    s = load_scene(name)
    pt = ma.PathTracing(s, beta=2.0)
    xy, si = _paths(64, 48, 12000, 23)
    monkeypatch.setenv("MI_PT_WIDE_NODES", str(wide))
    monkeypatch.setenv("MI_BPT_PERSIST", "0")
    a = pt.bpt_trace_paths(64, 48, xy, si, seed=8)
    img_a = pt.bpt_render_rgbn(40, 30, spp=5, seed=3)
    monkeypatch.setenv("MI_BPT_PERSIST", "1")
    b = pt.bpt_trace_paths(64, 48, xy, si, seed=8)
    img_b = pt.bpt_render_rgbn(40, 30, spp=5, seed=3)
    assert np.array_equal(a[2], b[2]) and _bits_equal(a[0], b[0]).all() and _bits_equal(a[1], b[1]).all()
    assert np.array_equal(img_a[..., 3], img_b[..., 3])
    np.testing.assert_allclose(img_a, img_b, rtol=2e-6, atol=0)  # FP64 splat order is free
    if name in ("CornellBoxSpecular", "MetalRings"):  # and against the oracle directly
        o = oracle.Oracle(s, beta=2.0).bpt_trace_paths(64, 48, xy[:3000], si[:3000], seed=8)
        assert np.array_equal(np.asarray(b[2])[:3000], np.asarray(o[2])) and _bits_equal(b[0][:3000], o[0]).all() and _bits_equal(b[1][:3000], o[1]).all()


@pytest.mark.parametrize("name", ["CornellBoxSpecular", "LivingRoomLit"])
def test_bpt_launches_in_flight_render_the_same_image(monkeypatch, name):
    """r04: mi_bpt_render deals the paths of a launch to launches in flight on streams of their own (trace -> item count -> connect each; one commit per
    batch of frames).  Every path is what it is whichever launch holds it: denominators equal, sums equal up to the free FP64 order of the splats."""
    s = load_scene(name)
    pt = ma.PathTracing(s, beta=2.0)
    imgs = []
    for fl in ("1", "2", "4"):
        monkeypatch.setenv("MI_BPT_FLIGHTS", fl)
        imgs.append(pt.bpt_render_rgbn(256, 192, spp=12, seed=4))  # 590 k paths: several launches per flight
        assert pt.last_stats.num_paths == 256 * 192 * 12
    for b in imgs[1:]:
        assert np.array_equal(imgs[0][..., 3], b[..., 3])
        np.testing.assert_allclose(imgs[0], b, rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("set_aside", ["1", "0"])
def test_bpt_overflow_while_launches_are_in_flight(monkeypatch, set_aside):
    """sub-paths that outgrow their slab share while two launches are in flight.  Default: the paths are set aside and traced again at 1024 vertices when the
    batch's launches are done.  MI_BPT_SET_ASIDE=0: the shared counter does not say whose the overflow was — every trace not yet connected is redone one at a
    time in slices at a larger share (bpt_launch).  Closed furnace at roulette 0.97 (sub-paths of a hundred vertices) with a slab budget that leaves 16
    vertices: the image equals the one rendered with room for every path."""
    s = load_scene("TestCaseFurnace")
    pt = ma.PathTracing(s, beta=2.0, roulette=0.97)
    monkeypatch.setenv("MI_BPT_FLIGHTS", "2"); monkeypatch.setenv("MI_BPT_SET_ASIDE", set_aside)
    ref = pt.bpt_render_rgbn(256, 256, spp=5, seed=11)
    monkeypatch.setenv("MI_BPT_SLAB_MB", "64")
    a = pt.bpt_render_rgbn(256, 256, spp=5, seed=11)
    assert np.array_equal(ref[..., 3], a[..., 3])
    np.testing.assert_allclose(ref, a, rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("name,slab_mb", [("CornellBoxSpecular", "600"), ("LivingRoomLit", "")])
def test_bpt_paths_set_aside_for_the_launch_at_full_capacity(monkeypatch, name, slab_mb):
    """r04: a path whose sub-path outgrows the slab share of its launch is set aside (its index appended to a list, nothing of it counted) and traced again,
    with the others of its batch of frames, in a launch at the reference's 1024 vertices (BPT.hpp:30) before the frames are committed — so the share can be short
    (80 vertices hold all but 0.02 % of the sub-paths at roulette 0.9) and a launch holds four times the paths.  Same image as with every launch redone in
    slices (MI_BPT_SET_ASIDE=0): with a 600 MB budget (20 vertices: one path in eight is set aside) and at the default size."""
    s = load_scene(name)
    pt = ma.PathTracing(s, beta=2.0)
    if slab_mb: monkeypatch.setenv("MI_BPT_SLAB_MB", slab_mb)
    monkeypatch.setenv("MI_BPT_SET_ASIDE", "0")
    a = pt.bpt_render_rgbn(256, 192, spp=12, seed=9)
    st_a = pt.last_stats
    monkeypatch.setenv("MI_BPT_SET_ASIDE", "1")
    b = pt.bpt_render_rgbn(256, 192, spp=12, seed=9)
    st_b = pt.last_stats
    assert np.array_equal(a[..., 3], b[..., 3])
    np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-12)
    assert (st_a.num_paths, st_a.num_basic_rays, st_a.num_shadow_rays) == (st_b.num_paths, st_b.num_basic_rays, st_b.num_shadow_rays)


def test_bpt_visibility_stage_against_the_oracle(monkeypatch):
    """the forced visibility stage against the CPU oracle directly (not only against the other device form)"""
    monkeypatch.setenv("MI_BPT_DYN_VIS", "1")
    s = load_scene("CornellBoxSpecular")
    pt = ma.PathTracing(s, beta=2.0); orc = oracle.Oracle(s, beta=2.0)
    xy, si = _paths(48, 32, 3000, 3)
    a = pt.bpt_trace_paths(48, 32, xy, si, seed=9)
    b = orc.bpt_trace_paths(48, 32, xy, si, seed=9)
    assert np.array_equal(np.asarray(a[2]), np.asarray(b[2])) and _bits_equal(a[0], b[0]).all() and _bits_equal(a[1], b[1]).all()


def test_bpt_long_paths_fall_back_to_the_reference_capacity(monkeypatch):
    """roulette 0.97 in a closed furnace: sub-paths of a hundred vertices.  With a tiny slab budget the first attempt overflows and
    the launch is redone in slices at 1024 vertices per sub-path (BPT.hpp:30) — same bits as the oracle."""
    s = load_scene("TestCaseFurnace")
    monkeypatch.setenv("MI_BPT_SLAB_MB", "4")
    pt, orc = ma.PathTracing(s, beta=2.0, roulette=0.97), oracle.Oracle(s, beta=2.0, roulette=0.97)
    xy, si = _paths(32, 32, 1500, 2)
    gr, gs, gc = pt.bpt_trace_paths(32, 32, xy, si, seed=4); orr, os_, oc = orc.bpt_trace_paths(32, 32, xy, si, seed=4)
    assert gc[:, 0].max() > 100  # long paths indeed
    assert np.array_equal(gc, oc) and _bits_equal(gr, orr).all() and _bits_equal(gs, os_).all()


def test_bpt_sky_gradient():
    """--sky-horizon / --sky-zenith (Options.cpp:74-76): camera rays that leave the scene return the gradient times 1/roulette."""
    s = load_scene("TestCase0")  # a lit plane under an open sky
    pt, orc = ma.PathTracing(s, beta=2.0), oracle.Oracle(s, beta=2.0)
    pt.bpt_set_sky((0.5, 0.25, 0.125), (0.0, 0.0, 2.0)); orc.bpt_set_sky((0.5, 0.25, 0.125), (0.0, 0.0, 2.0))
    xy, si = _paths(64, 48, 6000, 3)
    gr, gs, gc = pt.bpt_trace_paths(64, 48, xy, si, seed=1); orr, os_, oc = orc.bpt_trace_paths(64, 48, xy, si, seed=1)
    assert np.array_equal(gc, oc) and _bits_equal(gr, orr).all() and _bits_equal(gs, os_).all()
    assert (gr[:, 2] != gr[:, 0]).any()  # the sky is visible
    img = pt.bpt_render_rgbn(48, 36, spp=8, seed=2); ref = orc.bpt_render_rgbn(48, 36, spp=8, seed=2, threads=8)
    np.testing.assert_allclose(img, ref, rtol=2e-6, atol=0)


def test_bpt_error_behaviour(cornell):
    pt = ma.PathTracing(cornell)
    with pytest.raises(ma.MiError):
        pt.bpt_render_rgbn(16, 16, spp=0)
    with pytest.raises(ma.MiError):
        pt.bpt_trace_paths(16, 16, np.array([[16, 0]], np.uint32), np.array([0], np.uint64))
    with pytest.raises(ma.MiError):
        pt.bpt_render_rgbn(16, 16, spp=1, camera_id=5)


def test_bpt_rejects_images_wider_than_its_pixel_packing_and_survives_a_tiny_slab_budget(monkeypatch):
    """The staged BPT packs a path's pixel as (y << 16) | x: wider / taller images are refused, not rendered into wrong pixels.  The vertex
    slabs are sized from the memory that is free (40 %, at most 3 x 16 GB) and shrink on an allocation failure: with a budget of 1 MB
    (MI_BPT_SLAB_MB) the frame still renders, identical per path (long sub-paths go through the slice path)."""
    s = load_scene("CornellBoxDiffuse")
    pt = ma.PathTracing(s, roulette=0.9, beta=2.0)
    with pytest.raises(ma.MiError) as e:
        pt.bpt_render_rgbn(70000, 1, spp=1, seed=1)
    assert e.value.code == -5 and "65535" in str(e.value)
    xy = np.stack(np.meshgrid(np.arange(24), np.arange(20)), -1).reshape(-1, 2).astype(np.uint32)
    si = np.zeros(len(xy), np.uint64)
    ref = pt.bpt_trace_paths(24, 20, xy, si, seed=5)
    monkeypatch.setenv("MI_BPT_SLAB_MB", "1")
    small = ma.PathTracing(s, roulette=0.9, beta=2.0).bpt_trace_paths(24, 20, xy, si, seed=5)
    for a, b in zip(ref, small):
        assert np.array_equal(a, b)
