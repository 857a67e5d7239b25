"""Parity tests proper (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs.

Bars: bit-exact for integer / index work (LBVH, primitive ids, ray counts, sample counts) and —
because oracle and device state the same arithmetic contract — bit-exact for the FP32 results of
the diffuse / mirror / dielectric paths too, and for Phong lobes and any beta as well: pow() is the
build's own definition (mi_powf: FP64 log2 / exp2 polynomials of explicit fmas), stated identically on both sides.  Full-size runs use size-independent properties
(sample counts, additivity over sample ranges, determinism, furnace value)."""
import os

import numpy as np
import pytest

import master_amd as ma
import oracle
from master_amd import scenegen as sb
from conftest import load_scene

pytestmark = pytest.mark.gpu


def rays(scene, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = scene.positions.min(0), scene.positions.max(0)
    o = np.zeros(n, ma.SURFACE_DTYPE)
    o["position"] = rng.uniform(lo, hi, (n, 3)); g = rng.normal(size=(n, 3)); o["gnormal"] = g / np.linalg.norm(g, axis=1, keepdims=True)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def grid_paths(w, h, spp):
    xy = np.stack(np.meshgrid(np.arange(w), np.arange(h)), -1).reshape(-1, 2).astype(np.uint32)
    return np.tile(xy, (spp, 1)), np.repeat(np.arange(spp, dtype=np.uint64), w * h)


SCENES = ["CornellBoxDiffuse", "CornellBoxSpecular", "TestCaseFurnace", "TestCase0", "DoubleLight", "soup300", "soup20000", "single",
          "MirrorBalls", "MetalRings", "LivingRoomLit"]  # the last three: real models with far-away lights / mixed triangle sizes


@pytest.fixture(params=["ploc", "lbvh"])
def builder(request, monkeypatch):
    """Both hierarchy builders; the product and the oracle read the same MI_PT_BVH switch at scene creation."""
    monkeypatch.setenv("MI_PT_BVH", request.param)
    return request.param


def get_scene(name):
    if name.startswith("soup"):
        return sb.random_soup(int(name[4:]), seed=11)
    if name == "single":  # one surface triangle + light: n_nodes = 1, exercises the tiny-tree paths
        b = sb.Builder()
        b.add_camera((0, -3, 0.5), (0, 1, 0))
        m = b.add_material(sb.material(ma.BSDF_DIFFUSE, diffuse=(0.5, 0.5, 0.5)))
        b.add_mesh([[(-1, 0, 0), (1, 0, 0), (0, 0, 1.5)]], m)
        b.add_light((0, -1, 2), (0, 0.5, -1), (0, 1, 0.5), (0.5, 0.5), (5, 5, 5))
        return b.build()
    return load_scene(name)


@pytest.mark.parametrize("name", SCENES)
def test_bvh_bit_exact(builder, name):
    s = get_scene(name)
    pt, orc = ma.PathTracing(s), oracle.Oracle(s)
    gi, oi = pt.bvh_info(), orc.bvh_info()
    assert (gi.n_nodes, gi.max_depth, gi.builder, gi.build_rounds) == (oi.n_nodes, oi.max_depth, oi.builder, oi.build_rounds)
    assert gi.builder == (1 if builder == "ploc" else 0)
    assert list(gi.scene_lo) == list(oi.scene_lo) and list(gi.scene_hi) == list(oi.scene_hi)
    gn, gs, gm = pt.bvh(); on, os_, om = orc.bvh()
    assert np.array_equal(gm, om) and np.array_equal(gs, os_)
    assert gn.tobytes() == on.tobytes()
    assert gi.stack_entries + 64 >= gi.max_depth - 1 and gi.stack_entries <= 12  # LDS rows + private-memory spill


@pytest.mark.parametrize("name", SCENES)
def test_intersect_and_occluded_bit_exact(builder, name):
    s = get_scene(name)
    pt, orc = ma.PathTracing(s), oracle.Oracle(s)
    o, d = rays(s, 100000, 1)
    gh, gt, gp = pt.intersect(o, d); oh, ot, op = orc.intersect(o, d)
    assert np.array_equal(gp, op) and np.array_equal(gt, ot) and gh.tobytes() == oh.tobytes()
    tg, _ = rays(s, 100000, 2)
    assert np.array_equal(pt.occluded(o, tg), orc.occluded(o, tg))
    # traversal == brute force (oracle without its BVH): exact t, exact primitive
    orc.set_use_bvh(False)
    nb = 20000 if s.n_triangles <= 20000 else 3000
    _, bt, bp = orc.intersect(o[:nb], d[:nb])
    assert np.array_equal(gp[:nb], bp) and np.array_equal(gt[:nb], bt)


@pytest.mark.parametrize("name", ["CornellBoxDiffuse", "CornellBoxSpecular", "MirrorBalls", "MetalRings", "LivingRoomLit", "soup20000", "single", "TestCase27"])
def test_wide_node_walk_is_exact(monkeypatch, name):
    """The HBM-resident kernels of large scenes walk 64-byte wide nodes (four grandchildren per record).  Forced on for
    small scenes here: closest hits, visibility and whole paths must not change (the hit is the (t, id) minimum)."""
    s = get_scene(name)
    monkeypatch.setenv("MI_PT_WIDE_NODES", "0"); binary = ma.PathTracing(s)
    monkeypatch.setenv("MI_PT_WIDE_NODES", "1"); pt = ma.PathTracing(s)
    orc = oracle.Oracle(s)
    pt.set_kernel(ma.KERNEL_MEGA_GLOBAL); binary.set_kernel(ma.KERNEL_MEGA_GLOBAL)
    o, d = rays(s, 100000, 7)
    gh, gt, gp = pt.intersect(o, d); oh, ot, op = orc.intersect(o, d)
    assert np.array_equal(gp, op) and np.array_equal(gt, ot) and gh.tobytes() == oh.tobytes()
    tg, _ = rays(s, 100000, 8)
    assert np.array_equal(pt.occluded(o, tg), orc.occluded(o, tg))
    # whole paths against the binary walk on the same device (same pow on both sides: exact also for Phong scenes)
    xy, si = grid_paths(48, 40, 6)
    g, gc = pt.trace_paths(48, 40, xy, si, seed=7); r, rc = binary.trace_paths(48, 40, xy, si, seed=7)
    assert np.array_equal(gc, rc)
    assert ((g.view(np.uint32) == r.view(np.uint32)) | (np.isnan(g) & np.isnan(r))).all()
    img = pt.render_rgbn(64, 48, spp=8, seed=5); ref = binary.render_rgbn(64, 48, spp=8, seed=5)
    assert np.array_equal(img, ref)


def test_intersect_edge_cases(cornell):
    pt, orc = ma.PathTracing(cornell), oracle.Oracle(cornell)
    o, d = rays(cornell, 64, 3)
    d[:16] = [0, 0, 1]; d[16:32] = [1, 0, 0]; d[32:40] = [0, -1, 0]       # axis-parallel: 1/0 slabs
    o["position"][40:48] = cornell.positions[cornell.indices[:8, 0]]          # start exactly on vertices
    o["position"][48:56] = [1e6, 1e6, 1e6]                                    # far outside
    gh, gt, gp = pt.intersect(o, d); oh, ot, op = orc.intersect(o, d)
    assert np.array_equal(gp, op) and np.array_equal(gt, ot) and gh.tobytes() == oh.tobytes()
    h, t, p = pt.intersect(o[:0], d[:0])                                       # empty batch
    assert len(h) == 0


@pytest.mark.parametrize("name,max_path", [("CornellBoxDiffuse", 8), ("CornellBoxDiffuse", 1), ("CornellBoxDiffuse", 2), ("CornellBoxDiffuse", 0),
                                           ("TestCaseFurnace", ma.PTRDIFF_MAX), ("CornellBoxSpecular", 12), ("DoubleLight", 6), ("single", 4)])
def test_per_path_radiance_bit_exact(name, max_path):
    s = get_scene(name)
    pt, orc = ma.PathTracing(s, max_path=max_path), oracle.Oracle(s, max_path=max_path)
    xy, si = grid_paths(48, 40, 6)
    gr, gc = pt.trace_paths(48, 40, xy, si, seed=7); orr, oc = orc.trace_paths(48, 40, xy, si, seed=7)
    assert np.array_equal(gc, oc)
    same = (gr.view(np.uint32) == orr.view(np.uint32)) | (np.isnan(gr) & np.isnan(orr))  # DoubleLight has Phong materials: own pow on both sides
    assert same.all(), "mismatching paths: %s" % np.nonzero(~same.all(1))[0][:10]


@pytest.mark.parametrize("beta,roulette,lights", [(2.0, 0.5, 1.0), (1.0, 1.0, 0.0), (1.5, 0.9, 1.0), (0.0, 0.3, 2.0)])
def test_parameters_follow_the_reference_semantics(cornell, beta, roulette, lights):
    pt = ma.PathTracing(cornell, max_path=6, beta=beta, roulette=roulette, lights=lights)
    orc = oracle.Oracle(cornell, max_path=6, beta=beta, roulette=roulette, lights=lights)
    xy, si = grid_paths(32, 32, 4)
    gr, gc = pt.trace_paths(32, 32, xy, si, seed=3); orr, oc = orc.trace_paths(32, 32, xy, si, seed=3)
    assert np.array_equal(gc, oc)
    assert np.array_equal(gr, orr)  # beta = 1.5 goes through mi_powf on both sides


@pytest.mark.parametrize("kernel", [ma.KERNEL_MEGA_LDS, ma.KERNEL_MEGA_GLOBAL, ma.KERNEL_WAVEFRONT])
@pytest.mark.parametrize("w,h,spp,window", [(64, 64, 8, None), (37, 23, 5, None), (64, 48, 3, (5, 7, 21, 30)), (8, 8, 1, None), (1, 1, 64, None), (130, 9, 2, (120, 0, 10, 9))])
def test_render_equals_oracle(cornell, kernel, w, h, spp, window):
    """Technique::render for spp frames: image, denominators and ray counters (ragged sizes, windows)."""
    pt, orc = ma.PathTracing(cornell, max_path=8), oracle.Oracle(cornell, max_path=8)
    pt.set_kernel(kernel)
    img = pt.render_rgbn(w, h, spp=spp, seed=5, sample_offset=3, window=window)
    ref = orc.render_rgbn(w, h, spp=spp, seed=5, sample_offset=3, window=window)
    st, so = pt.last_stats, orc.last_stats
    assert (st.num_paths, st.num_basic_rays, st.num_shadow_rays, st.numeric_errors) == (so.num_paths, so.num_basic_rays, so.num_shadow_rays, so.numeric_errors)
    assert np.array_equal(img[..., 3], ref[..., 3])
    # FP64 per-pixel sums are order independent up to FP64 rounding: after the FP32 cast at most 1 ulp
    np.testing.assert_allclose(img, ref, rtol=1.2e-7, atol=0)
    if window:
        x0, y0, ww, hh = window
        mask = np.ones((h, w), bool); mask[y0:y0 + hh, x0:x0 + ww] = False
        assert not img[mask].any()


def test_both_kernel_variants_agree_and_runs_are_deterministic(cornell):
    pt = ma.PathTracing(cornell, max_path=8)
    pt.set_kernel(ma.KERNEL_MEGA_LDS); a = pt.render_rgbn(96, 96, spp=32, seed=9); a2 = pt.render_rgbn(96, 96, spp=32, seed=9)
    pt.set_kernel(ma.KERNEL_MEGA_GLOBAL); b = pt.render_rgbn(96, 96, spp=32, seed=9)
    assert np.array_equal(a[..., 3], b[..., 3])
    np.testing.assert_allclose(a, a2, rtol=1.2e-7); np.testing.assert_allclose(a, b, rtol=1.2e-7)
    pt.set_kernel(ma.KERNEL_WAVEFRONT); w = pt.render_rgbn(96, 96, spp=32, seed=9); w2 = pt.render_rgbn(96, 96, spp=32, seed=9)
    assert pt.get_kernel() == ma.KERNEL_WAVEFRONT and np.array_equal(w, w2) and np.array_equal(a[..., 3], w[..., 3])
    np.testing.assert_allclose(a, w, rtol=1.2e-7)
    c = pt.render_rgbn(96, 96, spp=32, seed=10)
    assert not np.array_equal(a, c)


def test_numeric_errors_are_dropped_not_accumulated():
    """TransmissionBSDF has no TIR guard: sqrt(<0) = NaN; such samples are dropped and counted,
    and denom is NOT incremented (BSDF.cpp:480-493, Technique.cpp:222-230)."""
    s = load_scene("CornellBoxSpecular")
    pt, orc = ma.PathTracing(s, max_path=10), oracle.Oracle(s, max_path=10)
    img = pt.render_rgbn(64, 64, spp=16, seed=2); ref = orc.render_rgbn(64, 64, spp=16, seed=2)
    assert pt.last_stats.numeric_errors == orc.last_stats.numeric_errors
    assert np.isfinite(img).all()
    assert np.array_equal(img[..., 3], ref[..., 3])
    assert int((16 * 64 * 64) - img[..., 3].sum()) == pt.last_stats.numeric_errors
    np.testing.assert_allclose(img, ref, rtol=1.2e-7)


def test_sample_ranges_are_additive(cornell):
    """Streams are keyed on the global sample index: [0,24) == [0,8) + [8,24) (what the RCCL merge relies on)."""
    pt = ma.PathTracing(cornell, max_path=8)
    whole = pt.render_rgbn(64, 64, spp=24, seed=1, sample_offset=0)
    parts = pt.render_rgbn(64, 64, spp=8, seed=1, sample_offset=0).astype(np.float64) + pt.render_rgbn(64, 64, spp=16, seed=1, sample_offset=8)
    assert np.array_equal(whole[..., 3], parts[..., 3])
    np.testing.assert_allclose(whole, parts, rtol=3e-7)


def test_technique_render_accumulates_like_the_reference(cornell):
    """Technique::render adds one frame per call to the dvec4 view and fills statistics (Technique.cpp:47-76)."""
    pt = ma.PathTracing(cornell, max_path=4)
    view = np.zeros((32, 32, 4), np.float64)
    for i in range(3):
        rec = pt.render(view, seed=1)
        assert rec["sample_index"] == i and rec["numeric_errors"] == 0
    st = pt.statistics()
    assert st.num_samples == 3 and np.all(view[..., 3] == 3) and len(st.records) == 3
    orc = oracle.Oracle(cornell, max_path=4)
    ref = orc.render_rgbn(32, 32, spp=3, seed=1)
    np.testing.assert_allclose(view, ref, rtol=3e-7)
    assert (st.num_basic_rays, st.num_shadow_rays) == (orc.last_stats.num_basic_rays, orc.last_stats.num_shadow_rays)


def test_exr_checkpoint_carries_the_reference_metadata(cornell, tmp_path):
    """Application::_save (Application.cpp:251) -> save_exr: sums + denom + statistics/options as string attributes."""
    pt = ma.PathTracing(cornell, max_path=4)
    view = np.zeros((24, 24, 4), np.float64)
    pt.render(view, seed=1); pt.render(view, seed=1)
    meta = dict(pt.statistics().to_dict()); meta.update(pt.options_dict(24, 24))
    p = str(tmp_path / "c.exr")
    ma.save_exr(p, view.astype(np.float32), meta)
    raw = open(p, "rb").read()
    for key in (b"statistics.num_samples\x00string\x00", b"statistics.num_basic_rays\x00", b"records[1].frame_duration\x00", b"options.technique\x00string\x00", b"options.max_path\x00"):
        assert key in raw
    assert np.array_equal(ma.load_exr(p), view.astype(np.float32)) and meta["statistics.num_samples"] == "2" and meta["options.technique"] == "PT"


def test_white_furnace_on_device():
    s = load_scene("TestCaseFurnace")
    pt = ma.PathTracing(s)
    img = pt.render_rgbn(64, 64, spp=512, seed=3)
    rgb = img[..., :3] / img[..., 3:]
    assert rgb.mean() == pytest.approx(1.0, abs=0.002) and np.abs(rgb - 1).max() < 0.08


def test_full_size_workload_properties(cornell):
    """BASELINE configs[1] at full size: 512x512, 1024 spp, max path 8 — checked through size-independent
    properties: every pixel has denom == spp, path count exact, image statistically equal to the oracle."""
    pt = ma.PathTracing(cornell, max_path=8)
    img = pt.render_rgbn(512, 512, spp=1024, seed=0x5EED)
    st = pt.last_stats
    assert st.num_paths == 512 * 512 * 1024 and st.numeric_errors == 0
    assert np.all(img[..., 3] == 1024)
    assert st.num_basic_rays >= st.num_paths and st.num_shadow_rays <= st.num_basic_rays
    # BASELINE.md parity rule on a converged window of the frame: RMSE(GPU, CPU) <= 1.5 x RMSE(CPU, CPU') and mean-radiance bias < 0.5 %.
    # CPU side: the oracle's 64x64 crop at 1024 spp, two independent seeds, rendered in the build container (tests/tools/make_golden.py).
    from conftest import ROOT
    x0, y0, w, h = 224, 160, 64, 64
    a = np.load(os.path.join(ROOT, "tests", "golden", "c2_crop_1024spp_a.npy"))
    b = np.load(os.path.join(ROOT, "tests", "golden", "c2_crop_1024spp_b.npy"))
    assert np.all(a[..., 3] == 1024) and np.all(b[..., 3] == 1024)
    a, b = a[..., :3] / 1024, b[..., :3] / 1024
    gpu = img[y0:y0 + h, x0:x0 + w, :3] / 1024
    rmse = lambda x, y: float(np.sqrt(np.mean((x - y) ** 2)))
    assert rmse(gpu, a) <= 1.5 * rmse(a, b) and rmse(gpu, b) <= 1.5 * rmse(a, b)
    assert abs(gpu.mean() - 0.5 * (a.mean() + b.mean())) / gpu.mean() < 0.005
    # linearity in the sample range: two half renders merge to the same image
    half = pt.render_rgbn(512, 512, spp=512, seed=0x5EED, sample_offset=0).astype(np.float64) + pt.render_rgbn(512, 512, spp=512, seed=0x5EED, sample_offset=512)
    np.testing.assert_allclose(img, half, rtol=3e-7)


@pytest.mark.parametrize("label,spec,w,h,spp", [("C3'", "CornellBoxSpecular", 1024, 1024, 512), ("C4'", "atrium", 1920, 1080, 256)])
def test_full_size_stand_ins_properties(label, spec, w, h, spp):
    """BASELINE configs[2] and [3] at their full sizes on the stand-in scenes (the real .blend files are missing from the
    reference tree): unbounded paths; checked through size-independent properties — denominators + dropped samples = spp, exact path
    count, two half renders with disjoint sample ranges sum to the whole (linearity / merge_exr), a window of the frame against the
    oracle rendering that window."""
    s = load_scene(spec) if spec != "atrium" else sb.atrium()
    pt = ma.PathTracing(s)
    img = pt.render_rgbn(w, h, spp=spp, seed=0x5EED)
    st = pt.last_stats
    assert st.num_paths == w * h * spp
    assert int(w * h * spp - img[..., 3].astype(np.float64).sum()) == st.numeric_errors and np.isfinite(img).all()
    first = pt.render_rgbn(w, h, spp=spp // 2, seed=0x5EED, sample_offset=0); n1 = (pt.last_stats.num_basic_rays, pt.last_stats.num_shadow_rays)
    second = pt.render_rgbn(w, h, spp=spp // 2, seed=0x5EED, sample_offset=spp // 2); n2 = (pt.last_stats.num_basic_rays, pt.last_stats.num_shadow_rays)
    assert (n1[0] + n2[0], n1[1] + n2[1]) == (st.num_basic_rays, st.num_shadow_rays)
    assert np.array_equal(first[..., 3] + second[..., 3], img[..., 3])
    # three FP32 roundings of FP64 sums.  A mirror seen from behind its shading normal contributes with a negative sign (ReflectionBSDF: throughput
    # 1 / omega.y, BSDF.cpp:450-465), so the halves may cancel: the bound is relative to the parts, not to their sum
    f64, s64 = first.astype(np.float64), second.astype(np.float64)
    assert np.all(np.abs(f64 + s64 - img) <= 1.2e-7 * (np.abs(f64) + np.abs(s64) + np.abs(img)))
    win = (w // 2 - 16, h // 2 - 12, 32, 24)
    a = pt.render_rgbn(w, h, spp=2, seed=3, window=win); r = oracle.Oracle(s).render_rgbn(w, h, spp=2, seed=3, window=win)
    x0, y0, ww, hh = win
    assert np.array_equal(a[..., 3], r[..., 3])
    np.testing.assert_allclose(a, r, rtol=1.2e-7, atol=0)  # Phong scenes too: one FP32 rounding of the same FP64 sums


def test_large_procedural_scene_deep_tree():
    """C4' stand-in at test size (~60k triangles, BVH depth > 24, HBM-resident scene, Phong + mirror + glass):
    LBVH bit-exact, closest hit / shadow rays bit-exact, per-path radiance and ray counts bit-exact."""
    s = sb.atrium(60000)
    pt, orc = ma.PathTracing(s), oracle.Oracle(s)
    assert pt.get_kernel() == ma.KERNEL_MEGA_GLOBAL
    gi, oi = pt.bvh_info(), orc.bvh_info()
    assert (gi.n_nodes, gi.max_depth) == (oi.n_nodes, oi.max_depth) and gi.max_depth > 20
    gn, gs, gm = pt.bvh(); on, os_, om = orc.bvh()
    assert np.array_equal(gm, om) and np.array_equal(gs, os_) and gn.tobytes() == on.tobytes()
    o, d = rays(s, 50000, 4)
    gh, gt, gp = pt.intersect(o, d); oh, ot, op = orc.intersect(o, d)
    assert np.array_equal(gp, op) and np.array_equal(gt, ot) and gh.tobytes() == oh.tobytes()
    tg, _ = rays(s, 50000, 5)
    assert np.array_equal(pt.occluded(o, tg), orc.occluded(o, tg))
    rng = np.random.default_rng(1); n = 8000
    xy = np.stack([rng.integers(0, 320, n), rng.integers(0, 180, n)], 1).astype(np.uint32); si = rng.integers(0, 64, n).astype(np.uint64)
    g, gc = pt.trace_paths(320, 180, xy, si, seed=2); r, rc = orc.trace_paths(320, 180, xy, si, seed=2)
    assert np.array_equal(gc, rc)
    same = (g.view(np.uint32) == r.view(np.uint32)) | (np.isnan(g) & np.isnan(r))
    assert same.all(), "mismatching paths: %s" % np.nonzero(~same.all(1))[0][:10]
    # image level (BASELINE.md rule): RMSE(GPU, CPU) <= 1.5 x RMSE(CPU, CPU') at equal spp
    a = orc.render_rgbn(96, 54, spp=8, seed=1)[..., :3] / 8; b = orc.render_rgbn(96, 54, spp=8, seed=2)[..., :3] / 8
    gimg = pt.render_rgbn(96, 54, spp=8, seed=1)[..., :3] / 8
    rmse = lambda x, y: float(np.sqrt(np.nanmean((x - y) ** 2)))
    assert rmse(gimg, a) <= 1.5 * rmse(a, b)


def test_phong_scene_is_bit_exact():
    """CornellBoxPhong: every wall is a PhongBSDF; cos^n and the lobe sampling go through mi_powf on both sides."""
    s = load_scene("CornellBoxPhong")
    pt, orc = ma.PathTracing(s, max_path=6), oracle.Oracle(s, max_path=6)
    xy, si = grid_paths(40, 40, 4)
    g, gc = pt.trace_paths(40, 40, xy, si, seed=5); r, rc = orc.trace_paths(40, 40, xy, si, seed=5)
    assert np.array_equal(gc, rc)
    same = (g.view(np.uint32) == r.view(np.uint32)) | (np.isnan(g) & np.isnan(r))
    assert same.all(), "mismatching paths: %s" % np.nonzero(~same.all(1))[0][:10]
    img = pt.render_rgbn(64, 64, spp=64, seed=1); ref = orc.render_rgbn(64, 64, spp=64, seed=1)
    np.testing.assert_allclose(img, ref, rtol=1.2e-7, atol=0)


@pytest.mark.parametrize("name", ["CornellBoxDiffuse", "CornellBoxSpecular", "TestCaseFurnace", "MirrorBalls", "TestCase27", "MetalRings", "single"])
def test_wavefront_pipeline_is_bit_identical_per_path(name):
    """The wavefront kernels run the same device functions in the same order as the megakernel: radiance and ray counts
    of every path agree bit for bit (Phong included: same pow on both sides), also when a slot renders many paths in a row."""
    s = get_scene(name)
    a, b = ma.PathTracing(s), ma.PathTracing(s)
    b.set_kernel(ma.KERNEL_WAVEFRONT)
    rng = np.random.default_rng(23); n = 30000
    xy = np.stack([rng.integers(0, 96, n), rng.integers(0, 54, n)], 1).astype(np.uint32); si = rng.integers(0, 64, n).astype(np.uint64)
    ra, ca = a.trace_paths(96, 54, xy, si, seed=13); rb, cb = b.trace_paths(96, 54, xy, si, seed=13)
    assert np.array_equal(ca, cb)
    assert ((ra.view(np.uint32) == rb.view(np.uint32)) | (np.isnan(ra) & np.isnan(rb))).all()
    os.environ["MI_PT_WF_SLOTS"] = "1024"  # few slots: every slot renders ~30 paths one after the other
    try:
        rc, cc = b.trace_paths(96, 54, xy, si, seed=13)
    finally:
        del os.environ["MI_PT_WF_SLOTS"]
    assert np.array_equal(ca, cc) and ((ra.view(np.uint32) == rc.view(np.uint32)) | (np.isnan(ra) & np.isnan(rc))).all()
    ia = a.render_rgbn(160, 90, spp=24, seed=3); ib = b.render_rgbn(160, 90, spp=24, seed=3)
    assert np.array_equal(ia[..., 3], ib[..., 3])
    np.testing.assert_allclose(ia, ib, rtol=1.2e-7)
    sa, sb_ = a.last_stats, b.last_stats
    assert (sa.num_paths, sa.num_basic_rays, sa.num_shadow_rays, sa.numeric_errors) == (sb_.num_paths, sb_.num_basic_rays, sb_.num_shadow_rays, sb_.numeric_errors)


def _corpus():
    """Every .miscene fixture below 1 MB: the reference's own models (tools/make_scenes.py) — sun lights, glass of several
    IORs, mirrors, Phong, meshes from 4 to 5 332 triangles."""
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes")
    return sorted(f[:-8] for f in os.listdir(d) if f.endswith(".miscene") and os.path.getsize(os.path.join(d, f)) < 1000000)


@pytest.mark.parametrize("name", _corpus())
def test_reference_corpus_parity(name):
    """Tree, ray counts and per-path radiance against the oracle on every model of the reference that fits a fixture.
    Bit-exact on all of them, Phong scenes (material type 2) included."""
    s = load_scene(name)
    pt, orc = ma.PathTracing(s), oracle.Oracle(s)
    gn, gs, gm = pt.bvh(); on, os_, om = orc.bvh()
    assert np.array_equal(gm, om) and np.array_equal(gs, os_) and gn.tobytes() == on.tobytes()
    rng = np.random.default_rng(17); n = 6000
    xy = np.stack([rng.integers(0, 96, n), rng.integers(0, 54, n)], 1).astype(np.uint32); si = rng.integers(0, 32, n).astype(np.uint64)
    g, gc = pt.trace_paths(96, 54, xy, si, seed=11); r, rc = orc.trace_paths(96, 54, xy, si, seed=11)
    assert np.array_equal(gc, rc)
    same = (g.view(np.uint32) == r.view(np.uint32)) | (np.isnan(g) & np.isnan(r))
    assert same.all(), "mismatching paths: %s" % np.nonzero(~same.all(1))[0][:10]


def test_4k_frame_and_window_against_oracle(cornell):
    """BASELINE configs[4] resolution (3840x2160): index arithmetic at full size, and a 48x40 window of that frame
    against the oracle rendering the same window (pixel indices, camera rays and streams depend on the full resolution)."""
    pt, orc = ma.PathTracing(cornell, max_path=8), oracle.Oracle(cornell, max_path=8)
    img = pt.render_rgbn(3840, 2160, spp=2, seed=3)
    st = pt.last_stats
    assert st.num_paths == 3840 * 2160 * 2 and np.all(img[..., 3] == 2) and np.isfinite(img).all()
    win = (1901, 1003, 48, 40)
    ref = orc.render_rgbn(3840, 2160, spp=2, seed=3, window=win)
    x0, y0, w, h = win
    np.testing.assert_allclose(img[y0:y0 + h, x0:x0 + w], ref[y0:y0 + h, x0:x0 + w], rtol=1.2e-7)
    sub = pt.render_rgbn(3840, 2160, spp=2, seed=3, window=win)
    np.testing.assert_allclose(sub[y0:y0 + h, x0:x0 + w], ref[y0:y0 + h, x0:x0 + w], rtol=1.2e-7)
    assert pt.last_stats.num_paths == w * h * 2 == orc.last_stats.num_paths
    assert (pt.last_stats.num_basic_rays, pt.last_stats.num_shadow_rays) == (orc.last_stats.num_basic_rays, orc.last_stats.num_shadow_rays)


@pytest.mark.parametrize("kernel", [ma.KERNEL_MEGA_LDS, ma.KERNEL_MEGA_GLOBAL, ma.KERNEL_WAVEFRONT])
def test_pixel_tile_sharding_is_a_bitwise_partition_of_the_render(cornell, kernel):
    """mi_pt_set_tile_shard (BASELINE C5 / SURVEY 8(e)(ii)): rank r renders the 32x32 tiles {t : t mod R == r}, zeros
    elsewhere; each rank's framebuffer is the unsharded one under its mask, so the ranks' sum (merge_exr) IS the render."""
    from master_amd import dist as madist

    pt = ma.PathTracing(cornell, max_path=8)
    pt.set_kernel(kernel)
    for (W, H, win) in ((200, 136, None), (200, 136, (13, 7, 150, 100)), (24, 24, None)):
        full = pt.render_rgbn(W, H, spp=6, seed=9, sample_offset=4, window=win)
        full_stats = (pt.last_stats.num_paths, pt.last_stats.num_basic_rays, pt.last_stats.num_shadow_rays)
        for world in (2, 3, 8):
            own = madist.tile_owner(W, H, world, win)
            acc = np.zeros_like(full)
            stats = np.zeros(3, np.int64)
            for rank in range(world):
                pt.set_tile_shard(rank, world)
                fb = pt.render_rgbn(W, H, spp=6, seed=9, sample_offset=4, window=win)
                mine = own == rank
                assert np.array_equal(fb.view(np.uint32), np.where(mine[..., None], full, np.float32(0)).view(np.uint32)), (W, H, win, world, rank)
                assert pt.last_stats.num_paths == int(mine.sum()) * 6
                stats += (pt.last_stats.num_paths, pt.last_stats.num_basic_rays, pt.last_stats.num_shadow_rays)
                acc += fb
            assert np.array_equal(acc, full) and tuple(stats) == full_stats
        pt.set_tile_shard(0, 1)
        assert np.array_equal(pt.render_rgbn(W, H, spp=6, seed=9, sample_offset=4, window=win), full)
    with pytest.raises(ma.MiError):
        pt.set_tile_shard(2, 2)
    pt.set_tile_shard(1, 2)
    with pytest.raises(ma.MiError, match="sample ranges"):
        pt.bpt_render_rgbn(32, 32, spp=1)


@pytest.mark.parametrize("merge", ["device", "host"])
@pytest.mark.parametrize("kernel", [ma.KERNEL_AUTO, ma.KERNEL_MEGA_GLOBAL, ma.KERNEL_WAVEFRONT])
def test_one_process_multi_device_render_is_bit_identical(cornell, kernel, merge, monkeypatch):
    """mi_pt_render_multi: several handles (one per GPU; here all on this box's only GPU, on their own streams) render the
    tiles of one frame set concurrently; the merged framebuffer and the summed statistics equal one handle's render.  The frame is put together
    on the first handle's device (r03: a gather of every tile from its owner — peer reads over xGMI between GPUs — and ONE copy to the host) or,
    without peer access / with MI_PT_MULTI_HOST_MERGE=1, on the host from every device's framebuffer: same bits either way."""
    assert ma.device_count() >= 1
    if merge == "host":
        monkeypatch.setenv("MI_PT_MULTI_HOST_MERGE", "1")
    pts = [ma.PathTracing(cornell, max_path=8) for _ in range(3)]
    for t in pts:
        t.set_kernel(kernel)
    pts[1].set_tile_shard(1, 2)  # a shard set on a handle is ignored for the call and restored
    for (W, H, win, spp) in ((200, 136, None, 5), (200, 136, (13, 7, 150, 100), 1), (24, 24, None, 2)):
        ref = pts[0].render_rgbn(W, H, spp=spp, seed=11, sample_offset=3, window=win)
        rs = pts[0].last_stats
        for n in (1, 2, 3):
            img, st = ma.render_multi(pts[:n], W, H, spp=spp, seed=11, sample_offset=3, window=win)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (W, H, win, n)
            assert (st.num_paths, st.num_basic_rays, st.num_shadow_rays, st.numeric_errors) == (rs.num_paths, rs.num_basic_rays, rs.num_shadow_rays, rs.numeric_errors)
            assert kernel == ma.KERNEL_WAVEFRONT or st.gpu_ms > 0
            assert ma.lib().mi_pt_last_multi_merge() == (1 if merge == "device" else 0)
    # both merges on the SAME handles, back and forth (r03: the host path once released the device path's frame and event — a double free at destroy
    # whose error the next mi_pt_create reported as its own), then destroy and create again
    if merge == "device":
        ref2 = pts[0].render_rgbn(96, 64, spp=2, seed=5)
        for host in ("1", "0", "1", "0"):
            monkeypatch.setenv("MI_PT_MULTI_HOST_MERGE", host)
            img, _ = ma.render_multi(pts, 96, 64, spp=2, seed=5)
            assert np.array_equal(img.view(np.uint32), ref2.view(np.uint32)) and ma.lib().mi_pt_last_multi_merge() == (0 if host == "1" else 1)
        monkeypatch.delenv("MI_PT_MULTI_HOST_MERGE")
        spare = [ma.PathTracing(cornell, max_path=8) for _ in range(2)]
        ma.render_multi(spare, 64, 64, spp=1, seed=1)
        del spare
        again = ma.PathTracing(cornell, max_path=8)   # a release that failed would surface here
        assert np.array_equal(again.render_rgbn(24, 24, spp=2, seed=11, sample_offset=3), pts[0].render_rgbn(24, 24, spp=2, seed=11, sample_offset=3))
    own = pts[1].render_rgbn(64, 64, spp=1, seed=1)   # still sharded (1 of 2): the left 32 columns of the upper tile row... belong to rank 0
    assert (own[:32, :32, 3] == 0).all() and (own[:32, 32:, 3] == 1).all()
    with pytest.raises(ma.MiError):
        ma.render_multi([pts[0], pts[0]], 32, 32)


@pytest.mark.parametrize("name,beta", [("CornellBoxDiffuse", 1.0), ("CornellBoxDiffuse", 2.0), ("CornellBoxSpecular", 1.0), ("MirrorBalls", 2.0),
                                       ("DoubleLight", 1.0), ("CornellBoxDiffuse", 1.5), ("LivingRoomLit", 1.0), ("TestCase27", 1.0)])
def test_feature_specialised_kernels_render_the_same_image(monkeypatch, name, beta):
    """Scenes without Phong lobes / mirrors / glass / a general beta run megakernel variants compiled without that code
    (RenderParams::features); the image is the one the general variant renders, bit for bit, and the oracle's up to the FP32 cast."""
    s = get_scene(name)
    pt = ma.PathTracing(s, max_path=7, beta=beta)
    a = pt.render_rgbn(72, 56, spp=6, seed=4); sa = (pt.last_stats.num_basic_rays, pt.last_stats.num_shadow_rays)
    monkeypatch.setenv("MI_PT_PLAIN_KERNEL", "0")
    b = pt.render_rgbn(72, 56, spp=6, seed=4); sb_ = (pt.last_stats.num_basic_rays, pt.last_stats.num_shadow_rays)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa == sb_
    orc = oracle.Oracle(s, max_path=7, beta=beta)
    np.testing.assert_allclose(a, orc.render_rgbn(72, 56, spp=6, seed=4), rtol=1.2e-7, atol=0)
    assert sa == (orc.last_stats.num_basic_rays, orc.last_stats.num_shadow_rays)


def test_lossless_closed_scene_with_roulette_one_terminates():
    """roulette = 1 with unlimited path length in a closed box of albedo 1 never terminates in the reference; the device cuts a path after
    2^20 edges (mi_pt_params.max_path), so the launch ends — a kernel that never ends would take the GPU with it."""
    b = sb.Builder()
    b.add_camera((0, -0.9, 0), (0, 1, 0))
    m = b.add_material(sb.material(ma.BSDF_DIFFUSE, diffuse=(1.0, 1.0, 1.0)))
    q = [((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)), ((-1, -1, 1), (-1, 1, 1), (1, 1, 1), (1, -1, 1)),
         ((-1, -1, -1), (-1, -1, 1), (1, -1, 1), (1, -1, -1)), ((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1)),
         ((-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (-1, -1, 1)), ((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1))]
    for a, c, d, e in q:
        b.add_quad(a, c, d, e, m)
    b.add_light((0, 0, 0.5), (0, 0, -1), (0, 1, 0), (0.1, 0.1), (1, 1, 1))
    pt = ma.PathTracing(b.build(), roulette=1.0)  # max_path = PTRDIFF_MAX
    img = pt.render_rgbn(1, 1, spp=1, seed=2)
    st = pt.last_stats
    # ~10^6 segments: FP32 rounding of the albedo-1 throughput ends the walk (PT.cpp:62-64) or the cut does; either way the launch is bounded
    assert st.num_paths == 1 and 10000 < st.num_basic_rays < (1 << 22)
    assert img[0, 0, 3] in (0.0, 1.0)  # kept if its sum is finite


def test_degenerate_and_coincident_triangles():
    """Zero-area triangles are never hit (den == 0); coincident triangles tie on t and the smaller global index wins —
    on the device (BVH order) exactly as in the oracle (index order)."""
    b = sb.Builder()
    b.add_camera((0, -4, 1), (0, 1, 0))
    m0 = b.add_material(sb.material(ma.BSDF_DIFFUSE, diffuse=(0.8, 0.2, 0.2)))
    m1 = b.add_material(sb.material(ma.BSDF_DIFFUSE, diffuse=(0.2, 0.8, 0.2)))
    quad = [[(-1, 0, 0), (1, 0, 0), (1, 0, 2)], [(-1, 0, 0), (1, 0, 2), (-1, 0, 2)]]
    b.add_mesh(quad, m0)
    b.add_mesh(quad, m1)                                                   # coincident copy with another material
    b.add_mesh([[(0, 1, 0), (0, 1, 0), (0, 1, 0)], [(0, 1, 0), (1, 1, 0), (2, 1, 0)]], m0)   # point and line: zero area
    b.add_quad((-3, -3, 0), (3, -3, 0), (3, 3, 0), (-3, 3, 0), m1)
    b.add_light((0, -1, 3), (0, 0.3, -1), (0, 1, 0.3), (1, 1), (20, 20, 20))
    s = b.build()
    pt, orc = ma.PathTracing(s, max_path=5), oracle.Oracle(s, max_path=5)
    o, d = rays(s, 50000, 8)
    gh, gt, gp = pt.intersect(o, d); oh, ot, op = orc.intersect(o, d)
    assert np.array_equal(gp, op) and np.array_equal(gt, ot) and gh.tobytes() == oh.tobytes()
    assert not np.isin(gp, [4, 5]).any()                                    # the degenerate triangles
    assert not np.isin(gp, [2, 3]).any() and np.isin(gp, [0, 1]).any()      # of each coincident pair the lower index wins
    xy, si = grid_paths(32, 32, 4)
    g, gc = pt.trace_paths(32, 32, xy, si, seed=1); r, rc = orc.trace_paths(32, 32, xy, si, seed=1)
    assert np.array_equal(gc, rc) and np.array_equal(g, r)


def test_error_behaviour_on_device(cornell):
    pt = ma.PathTracing(cornell, max_path=8)
    with pytest.raises(ma.MiError) as e:
        pt.render_rgbn(32, 32, spp=1, camera_id=5)           # Cameras.cpp:48 runtime_assert(cameraId < size)
    assert e.value.code == -1
    with pytest.raises(ma.MiError):
        pt.render_rgbn(32, 32, spp=1, window=(30, 0, 8, 8))   # Technique.cpp:318 runtime_assert(xEnd <= width)
    with pytest.raises(ma.MiError):
        pt.render_rgbn(32, 32, spp=0)
    with pytest.raises(ma.MiError):
        ma.PathTracing(cornell, roulette=0.0)
    with pytest.raises(ma.MiError):
        ma.PathTracing(cornell, device=99)
    big = sb.random_soup(20000, seed=1)
    with pytest.raises(ma.MiError):
        ma.PathTracing(big).set_kernel(ma.KERNEL_MEGA_LDS)    # does not fit LDS


@pytest.mark.parametrize("wide", ["1", "0"])
def test_camera_far_outside_a_thin_scene_on_the_quantised_walks(monkeypatch, wide):
    """ADVICE r03 (medium).  The quantised node walks (wide and binary records, PT megakernel / traverse_dyn / BPT) test boxes on a 16-bit grid over the
    SCENE box with a fixed padding in cells and no per-ray slack.  A camera far outside a thin scene is millions of cells away on the thin axis
    (here 1 500 extents = 10^8 cells), where the roundings of the grid-space ray exceed one cell: with the one-cell padding of round 3 primary rays
    lose boxes and the image gets holes.  mi_pt_create now widens the padding with the distance of the farthest camera (1 cell per 2^20 cells), so
    every path equals the oracle's again; MI_PT_QUANT_PAD=1 restores the old padding and shows what the guard prevents."""
    import master_amd.scenegen as sg

    s = sg.thin_slab()
    monkeypatch.setenv("MI_PT_WIDE_NODES", wide)
    monkeypatch.setenv("MI_PT_FLOAT_NODES", "0")  # the median-triangle rule would pick the float nodes for this grid; the test is about the quantised ones
    xy, si = grid_paths(64, 48, 4)
    orr, oc = oracle.Oracle(s, max_path=5).trace_paths(64, 48, xy, si, seed=13)
    assert (oc[:, 0] >= 2).mean() > 0.9  # the camera looks at the slab: nearly every primary ray hits it and continues
    pt = ma.PathTracing(s, max_path=5)
    assert pt.get_kernel() == ma.KERNEL_MEGA_GLOBAL
    gr, gc = pt.trace_paths(64, 48, xy, si, seed=13)
    same = (gr.view(np.uint32) == orr.view(np.uint32)) | (np.isnan(gr) & np.isnan(orr))
    assert np.array_equal(gc, oc) and same.all(), "mismatching paths: %d" % int((~same.all(1)).sum())
    img = pt.render_rgbn(64, 48, spp=4, seed=13)
    assert pt.last_launch().wide_nodes == (1 if wide == "1" else 0)  # the quantised records, wide or binary
    ref = oracle.Oracle(s, max_path=5).render_rgbn(64, 48, spp=4, seed=13)
    np.testing.assert_allclose(img, ref, rtol=1.2e-7)
    # BPT walks the same records (eye sub-paths start at the camera)
    br, bs, bc = pt.bpt_trace_paths(64, 48, xy[:1024], si[:1024], seed=13)
    orb, osb, ocb = oracle.Oracle(s, max_path=5).bpt_trace_paths(64, 48, xy[:1024], si[:1024], seed=13)
    assert np.array_equal(bc, ocb) and np.array_equal(br.view(np.uint32), orb.view(np.uint32))
    # the old padding loses rays here (kept as evidence that the case is real, not as a requirement on the old behaviour)
    monkeypatch.setenv("MI_PT_QUANT_PAD", "1")
    old = ma.PathTracing(s, max_path=5)
    og, ogc = old.trace_paths(64, 48, xy, si, seed=13)
    lost = int((ogc[:, 0] != oc[:, 0]).sum())
    print("paths whose ray count changes with the one-cell padding: %d of %d" % (lost, len(oc)))


def test_nested_shells_deep_unbalanced_tree_on_the_wide_walk(monkeypatch):
    """r04: the wide records come from an area-guided collapse of the BVH2 (a record opens its child of largest area), so a record may span one to three BVH2
    levels and the stack capacity of the wide walk is the exact maximum found by the marking pass, not a depth formula.  A scene built to be deep and lopsided —
    sixty concentric octahedral shells, each 0.8 of the previous, with rays starting between the shells — walked through both node formats against the oracle."""
    rng = np.random.default_rng(4)
    b = sb.Builder()
    b.add_camera((0, -3, 0.2), (0, 1, 0))
    m = b.add_material(sb.material(ma.BSDF_DIFFUSE, diffuse=(0.6, 0.6, 0.6)))
    v = np.array([(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)], np.float64)
    faces = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    tris = []
    for k in range(60):
        r = 2.0 * 0.8 ** k
        for f in faces:
            if (k + f[0]) % 7 != 0:  # holes, so that rays thread several shells
                tris.append([v[i] * r + rng.normal(scale=1e-3 * r, size=3) for i in f])
    b.add_mesh(tris, m)
    b.add_light((0, 0, 2.6), (0, 0, -1), (0, 1, 0), (0.8, 0.8), (20, 20, 20))
    s = b.build()
    orc = oracle.Oracle(s)
    n = 20000
    o = np.zeros(n, ma.SURFACE_DTYPE)
    rad = 2.0 * 0.8 ** rng.uniform(0, 58, n)
    dirs = rng.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    o["position"] = (dirs * rad[:, None] * 0.7).astype(np.float32)
    o["gnormal"] = dirs.astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True); d = d.astype(np.float32)
    oh, ot, op = orc.intersect(o, d)
    assert (op != 0xFFFFFFFF).mean() > 0.5
    t = np.zeros(n, ma.SURFACE_DTYPE)
    t["position"] = (rng.normal(size=(n, 3)) * 0.5).astype(np.float32); t["gnormal"] = d
    ov = orc.occluded(o, t)
    xy, si = grid_paths(32, 24, 6)
    orr, oc = orc.trace_paths(32, 24, xy, si, seed=8)
    for wide in ("1", "0"):
        monkeypatch.setenv("MI_PT_WIDE_NODES", wide); monkeypatch.setenv("MI_PT_FLOAT_NODES", "0")
        pt = ma.PathTracing(s)
        assert pt.get_kernel() == ma.KERNEL_MEGA_GLOBAL and pt.bvh_info().max_depth >= 16
        gh, gt, gp = pt.intersect(o, d)
        assert np.array_equal(gp, op) and np.array_equal(gt, ot) and gh.tobytes() == oh.tobytes()
        assert np.array_equal(pt.occluded(o, t), ov)
        gr, gc = pt.trace_paths(32, 24, xy, si, seed=8)
        assert np.array_equal(gc, oc) and np.array_equal(gr.view(np.uint32), orr.view(np.uint32))
