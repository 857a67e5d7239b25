"""Pins the CPU oracle (CPU-only): closed-form results of the estimator it restates, the
reference's normalised test scenes, internal consistency (BVH == brute force, Embree semantics),
and frozen regression vectors (tests/golden)."""
import json
import math
import os

import numpy as np
import pytest

import master_amd as ma
import oracle
from master_amd import scenegen as sb
from conftest import ROOT, load_scene

GOLD = os.path.join(ROOT, "tests", "golden")


def test_rng_stream_matches_frozen_vectors():
    kat = json.load(open(os.path.join(GOLD, "rng_kat.json")))
    for c in kat["cases"]:
        v = oracle.rng_floats(c["seed"], c["pixel"], c["sample"], 8)
        assert [float(x).hex() for x in v] == c["floats_hex"]
        assert np.all((v >= 0) & (v < 1))


def test_rng_uniformity_and_independence_across_paths():
    v = np.concatenate([oracle.rng_floats(1, p, s, 4) for p in range(64) for s in range(64)])
    assert abs(v.mean() - 0.5) < 0.01 and abs(v.var() - 1 / 12) < 0.005
    a = np.array([oracle.rng_floats(1, p, 0, 1)[0] for p in range(4096)])
    b = np.array([oracle.rng_floats(1, p, 1, 1)[0] for p in range(4096)])
    assert abs(np.corrcoef(a, b)[0, 1]) < 0.05


def test_cornell_lbvh_and_paths_regression_pins(cornell):
    pin = json.load(open(os.path.join(GOLD, "cornell_lbvh_pin.json")))
    nodes, order, morton = oracle.Oracle(cornell).bvh()
    assert order.tolist() == pin["sorted_tri"] and morton.tolist() == pin["morton"]
    assert [[int(n["link0"]), int(n["link1"])] for n in nodes] == pin["links"]
    pins = json.load(open(os.path.join(GOLD, "cornell_paths_pin.json")))
    xy = np.stack(np.meshgrid(np.arange(32), np.arange(32)), -1).reshape(-1, 2).astype(np.uint32)
    xy = np.tile(xy, (4, 1)); si = np.repeat(np.arange(4, dtype=np.uint64), 1024)
    for c in pins["cases"]:
        rad, cnt = oracle.Oracle(cornell, max_path=c["max_path"]).trace_paths(32, 32, xy, si, seed=7)
        assert int(np.bitwise_xor.reduce(rad.view(np.uint32).ravel())) == c["xor_bits"]
        assert int(cnt[:, 0].sum()) == c["basic"] and int(cnt[:, 1].sum()) == c["shadow"]


def _flat_grid(nu, nv):
    """nu x nv identical quads in a plane: every neighbour pair has the same union area (PLOC tie-breaking)."""
    b = sb.FastBuilder()
    b.add_camera((0, -3, 2), (0, 1, -0.5))
    m = b.add_material(sb.material(ma.BSDF_DIFFUSE, diffuse=(0.5, 0.5, 0.5)))
    tris, nrm = sb.grid(nu, nv, lambda u, v: (np.stack([u * 4 - 2, v * 4 - 2, 0 * u], -1), np.stack([0 * u, 0 * u, 0 * u + 1], -1)))
    b.add_tris(tris, nrm, m)
    b.add_light((0, 0, 3), (0, 0, -1), (0, 1, 0), (1, 1), (5, 5, 5))
    return b.build()


@pytest.fixture(params=["ploc", "lbvh"])
def builder(request, monkeypatch):
    """Both hierarchy builders (the product and the oracle read the same MI_PT_BVH switch)."""
    monkeypatch.setenv("MI_PT_BVH", request.param)
    return request.param


@pytest.mark.parametrize("kind", ["soup", "grid"])
def test_bvh_is_a_valid_hierarchy(builder, kind):
    s = sb.random_soup(3000, seed=3) if kind == "soup" else _flat_grid(48, 48)
    o = oracle.Oracle(s)
    info = o.bvh_info()
    assert info.builder == (1 if builder == "ploc" else 0)
    if builder == "ploc":  # a constant fraction of the clusters merges per round, also when all areas tie
        assert 0 < info.build_rounds <= 8 * int(np.ceil(np.log2(s.n_triangles)))
    nodes, order, morton = o.bvh()
    n = s.n_triangles
    assert sorted(order.tolist()) == list(range(n)) and np.all(np.diff(morton.astype(np.int64)) >= 0)
    seen_leaf, seen_node = np.zeros(n, bool), np.zeros(n - 1, bool)
    seen_node[0] = True
    tri_lo = s.positions[s.indices].min(1); tri_hi = s.positions[s.indices].max(1)
    for i, nd in enumerate(nodes):
        for link, lo, hi in ((nd["link0"], nd["lo0"], nd["hi0"]), (nd["link1"], nd["lo1"], nd["hi1"])):
            if link < 0:
                assert not seen_leaf[~link]; seen_leaf[~link] = True
                t = order[~link]
                assert np.all(lo <= tri_lo[t]) and np.all(hi >= tri_hi[t])  # padded leaf box contains the triangle
            else:
                assert not seen_node[link]; seen_node[link] = True
                assert nodes[link]["parent"] == i
                clo = np.minimum(nodes[link]["lo0"], nodes[link]["lo1"]); chi = np.maximum(nodes[link]["hi0"], nodes[link]["hi1"])
                assert np.array_equal(lo, clo) and np.array_equal(hi, chi)
    assert seen_leaf.all() and seen_node.all()


def _rays(scene, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = scene.positions.min(0), scene.positions.max(0)
    o = np.zeros(n, ma.SURFACE_DTYPE)
    o["position"] = rng.uniform(lo, hi, (n, 3)); g = rng.normal(size=(n, 3)); o["gnormal"] = g / np.linalg.norm(g, axis=1, keepdims=True)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


@pytest.mark.parametrize("scene_name", ["CornellBoxDiffuse", "CornellBoxSpecular", "MirrorBalls", "soup"])
def test_bvh_traversal_equals_brute_force(builder, scene_name):
    s = sb.random_soup(2000, seed=5) if scene_name == "soup" else load_scene(scene_name)
    o = oracle.Oracle(s, use_bvh=True)
    org, d = _rays(s, 20000, 1)
    h1, t1, p1 = o.intersect(org, d)
    tg, _ = _rays(s, 20000, 2)
    v1 = o.occluded(org, tg)
    o.set_use_bvh(False)
    h2, t2, p2 = o.intersect(org, d)
    v2 = o.occluded(org, tg)
    assert np.array_equal(p1, p2) and np.array_equal(t1, t2) and h1.tobytes() == h2.tobytes() and np.array_equal(v1, v2)


def test_embree_hit_semantics(cornell):
    """P = (1-u-v) v0 + u v1 + v v2 on the reported triangle; gnormal faces the ray; frames orthonormal."""
    o = oracle.Oracle(cornell)
    org, d = _rays(cornell, 5000, 3)
    hits, t, prim = o.intersect(org, d)
    ok = prim != ma.UINT32_MAX
    assert ok.mean() > 0.5
    tri = cornell.positions[cornell.indices[prim[ok]]]
    p = hits["position"][ok]
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]); n /= np.linalg.norm(n, axis=1, keepdims=True)
    assert np.abs(np.einsum("ij,ij->i", p - tri[:, 0], n)).max() < 1e-5            # on the triangle's plane
    assert np.all(np.einsum("ij,ij->i", hits["gnormal"][ok], -d[ok]) >= 0)         # Scene.cpp:119-120
    T = hits["tangent"][ok].reshape(-1, 3, 3)
    gram = np.einsum("nij,nkj->nik", T, T)
    assert np.abs(gram - np.eye(3)).max() < 1e-5                                  # Gram–Schmidt (Scene.cpp:98-111)
    assert np.array_equal(hits["material_id"][ok], cornell.tri_material[prim[ok]])


def test_shadow_rays_ignore_light_quads(cornell):
    """Scene.cpp:42,173: ray mask 1<<mesh never matches the light geometry (mask 1<<light)."""
    l = cornell.lights[0]
    a = np.zeros(1, ma.SURFACE_DTYPE); b = np.zeros(1, ma.SURFACE_DTYPE)
    a["position"] = [[l.position[0], l.position[1], l.position[2] + 0.005]]  # just above the light quad (below the ceiling)
    a["gnormal"] = [[0, 0, -1]]
    b["position"] = [[l.position[0], l.position[1], 0.6]]; b["gnormal"] = [[0, 0, 1]]
    o = oracle.Oracle(cornell)
    assert o.occluded(a, b)[0] == 1.0                   # the light quad lies between the two points and does not block
    hits, t, prim = o.intersect(a, np.array([[0, 0, -1]], np.float32))
    assert (hits["material_id"][0] & 3) == ma.ENTITY_LIGHT  # ...but a closest-hit ray does see it (mask 0xFFFFFFFF, Scene.cpp:196)


def test_direct_lighting_matches_lamberts_formula():
    """max_path = 2 (direct light only): pixel radiance = rho/pi * E, E from the closed-form
    irradiance of a rectangular Lambertian emitter (NEE + BSDF hits, MIS-combined: PT.cpp:41,70-79)."""
    b = sb.Builder()
    b.add_camera((0.3, -2.0, 1.0), (0.1, 2.0, -1.0), up=(0, 0, 1), fovx=0.5)
    rho = (0.6, 0.5, 0.4)
    m = b.add_material(sb.material(ma.BSDF_DIFFUSE, diffuse=rho))
    b.add_quad((-5, -5, 0), (5, -5, 0), (5, 5, 0), (-5, 5, 0), m)
    exit_ = np.array([9.0, 6.0, 3.0])
    b.add_light((0.2, 0.4, 1.5), (0, 0, -1), (0, 1, 0), (1.0, 0.6), exit_)
    s = b.build()
    o = oracle.Oracle(s, max_path=2)
    W = H = 101
    win = (50, 50, 1, 1)
    img = o.render_rgbn(W, H, spp=40000, seed=5, window=win, threads=1)
    px = img[50, 50, :3] / img[50, 50, 3]
    # where does the pixel centre hit the floor?
    f = oracle.camera_setup(s.cameras[0], 1.0)
    m3 = np.array(list(f.view_to_world), np.float64).reshape(3, 3).T
    d = m3 @ oracle.ray_direction(50.5, 50.5, W, H, f.focal_length_y).astype(np.float64)
    cam = np.array(list(f.position), np.float64)
    x = cam + d * (-cam[2] / d[2])
    l = s.lights[0]
    t0, t2 = np.array(l.tangent[0:3]), np.array(l.tangent[6:9]); p = np.array(l.position[:])
    corners = [p - 0.5 * t0 - 0.3 * t2, p + 0.5 * t0 - 0.3 * t2, p + 0.5 * t0 + 0.3 * t2, p - 0.5 * t0 + 0.3 * t2]
    for c in range(3):
        expect = rho[c] / math.pi * sb.rect_irradiance(x, (0, 0, 1), corners, exit_[c] / math.pi)
        assert px[c] == pytest.approx(expect, rel=0.01)


@pytest.mark.parametrize("beta", [0.0, 1.0, 2.0, 1.5])
def test_mis_weights_are_unbiased_for_any_beta(beta):
    """The power heuristic with any exponent gives the same expectation (PT.cpp:72-74,113-115)."""
    s = load_scene("TestCase0")
    img = oracle.Oracle(s, beta=beta, max_path=2).render_rgbn(32, 32, spp=256, seed=9)
    assert (img[..., :3] / img[..., 3:]).mean() == pytest.approx(1.0, abs=0.015)  # TestCase0 is one plane: direct light is all there is


def test_white_furnace():
    """Closed box, albedo 0.5, every face covered by a one-sided emitter of radiance 0.5: L = Le / (1 - rho) = 1
    for every pixel — exercises light pass-through, NEE/MIS, roulette compensation, frames on all six faces."""
    s = load_scene("TestCaseFurnace")
    img = oracle.Oracle(s).render_rgbn(32, 32, spp=256, seed=2)
    rgb = img[..., :3] / img[..., 3:]
    assert rgb.mean() == pytest.approx(1.0, abs=0.004)
    assert np.abs(rgb - 1.0).max() < 0.1
    # truncating the path length truncates the Neumann series: sum_{k<K} Le rho^k
    for mp, expect in ((1, 0.5), (2, 0.75), (3, 0.875)):
        img = oracle.Oracle(s, max_path=mp).render_rgbn(16, 16, spp=64, seed=2)
        assert (img[..., :3] / img[..., 3:]).mean() == pytest.approx(expect, abs=0.01)


@pytest.mark.parametrize("name", ["TestCase0", "TestCase2", "TestCase5", "TestCase25", "TestCase3"])
def test_reference_test_scenes_average_to_one(name):
    """models/TestCase*.blend are normalised by the reference's author to an image average of 1 at the default
    512x512 aspect (tests/golden/reference_constants.json) — checks importer conventions + estimator together."""
    s = load_scene(name)
    img = oracle.Oracle(s).render_rgbn(64, 64, spp=256, seed=4)
    assert (img[..., :3] / img[..., 3:]).mean() == pytest.approx(1.0, abs=0.012)


def test_lights_scale_and_no_lights(cornell):
    """--no-lights (lights = 0) removes only directly visible emitters (PT.cpp:24), nothing else."""
    a = oracle.Oracle(cornell, max_path=3, lights=1.0).render_rgbn(32, 32, spp=8, seed=1)
    b = oracle.Oracle(cornell, max_path=3, lights=0.0).render_rgbn(32, 32, spp=8, seed=1)
    diff = (a - b)[..., :3]
    assert np.count_nonzero(diff.sum(-1)) < 0.1 * 32 * 32 and diff.max() > 1.0  # only the lamp's pixels differ


def test_bsdf_sampling_densities_integrate_to_one(cornell):
    """Diffuse: E[f cos / p] = albedo; Phong: sampled density matches the queried density."""
    s = load_scene("CornellBoxPhong")
    o = oracle.Oracle(s)
    phong = [i for i, m in enumerate(s.materials) if m.type == ma.BSDF_PHONG][0]
    diff = [i for i, m in enumerate(s.materials) if m.type == ma.BSDF_DIFFUSE][0]
    sp = np.zeros(1, ma.SURFACE_DTYPE)
    sp["gnormal"] = [[0, 0, 1]]; sp["tangent"] = [[1, 0, 0, 0, 0, 1, 0, 1, 0]]
    omega = np.array([0.3, 0.2, 0.93]); omega /= np.linalg.norm(omega)
    for mat in (diff, phong):
        sp["material_id"] = (mat << 2) | 1
        est = np.zeros(3)
        n = 4000
        for k in range(n):
            om, tp, d, dr, fin = o.bsdf_sample(sp, omega, seed=3, pixel=mat, sample=k)
            if d == 0.0:  # lobe sample below the horizon: throughput 0, the path ends at PT.cpp:62-64
                assert not np.any(tp)
                continue
            tq, dq, drq, _ = o.bsdf_query(sp, omega, om)
            assert dq == pytest.approx(d, rel=1e-5) and np.allclose(tq, tp, rtol=1e-5)
            est += tp * abs(om[2]) / d
        m = s.materials[mat]
        albedo = np.array(m.diffuse[:]) + (np.array(m.specular[:]) if m.type == ma.BSDF_PHONG else 0)
        assert np.all(est / n <= albedo * 1.05 + 1e-3)
        if m.type == ma.BSDF_DIFFUSE:
            np.testing.assert_allclose(est / n, albedo, rtol=1e-4)


def test_light_sampling_is_power_proportional():
    s = load_scene("DoubleLight")
    o = oracle.Oracle(s)
    ids = [o.light_sample(seed=1, pixel=0, sample=k)[0]["material_id"] for k in range(4000)]
    power = np.array([l.size[0] * l.size[1] * sum(abs(x) for x in l.exitance) for l in s.lights])
    for i, l in enumerate(s.lights):
        assert np.mean(np.array(ids) == l.material_id) == pytest.approx(power[i] / power.sum(), abs=0.03)


def test_rms_abs_errors_definition():
    rng = np.random.default_rng(0)
    rgbn = rng.uniform(0.5, 2, (7, 5, 4)).astype(np.float32); ref = rng.uniform(0, 1, (7, 5, 3)).astype(np.float32)
    d = np.abs(rgbn[..., :3] / rgbn[..., 3:] - ref)
    for fn in (ma.rms_abs_errors, oracle.rms_abs_errors):
        rms, ab = fn(rgbn, ref)
        assert rms == pytest.approx(math.sqrt((d ** 2).sum() / d.size), rel=1e-5) and ab == pytest.approx(d.sum() / d.size, rel=1e-5)
    # the entry point that takes the dvec4 view itself (ImageView.cpp:60-85 takes image_view_t<dvec4>): same definition; equal bits when the view holds float values
    rms_v, ab_v = ma.rms_abs_errors_view(rgbn.astype(np.float64), ref)
    assert (rms_v, ab_v) == ma.rms_abs_errors(rgbn, ref)
    view = rng.uniform(0.5, 2, (7, 5, 4)) * 1000.0  # double sums that are not floats
    dv = np.abs((view[..., :3] / view[..., 3:]).astype(np.float32) - ref)
    rms_v, ab_v = ma.rms_abs_errors_view(view, ref)
    assert rms_v == pytest.approx(math.sqrt((dv.astype(np.float64) ** 2).sum() / dv.size), rel=1e-5) and ab_v == pytest.approx(dv.sum() / dv.size, rel=1e-5)


def test_oracle_runs_the_whole_reference_corpus():
    """Every model fixture: finite radiance, at least the camera segment per path, no shadow rays where every light is a
    sun (PT.cpp:105-107 returns before Scene::occluded when the emitter's BSDF throughput is zero)."""
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes")
    names = sorted(f[:-8] for f in os.listdir(d) if f.endswith(".miscene") and os.path.getsize(os.path.join(d, f)) < 1000000)
    assert len(names) >= 55
    rng = np.random.default_rng(1); k = 800
    xy = np.stack([rng.integers(0, 64, k), rng.integers(0, 36, k)], 1).astype(np.uint32); si = rng.integers(0, 16, k).astype(np.uint64)
    for n in names:
        s = load_scene(n)
        rad, cnt = oracle.Oracle(s).trace_paths(64, 36, xy, si, seed=5)
        assert np.isfinite(rad).all() and (cnt[:, 0] >= 1).all(), n
        if all(l.diffuse == 0 for l in s.lights):
            assert cnt[:, 1].sum() == 0, n


def test_own_powf_is_within_one_ulp_and_follows_c99_special_cases():
    """mi_powf (Phong lobe, variable beta) is the build's own definition so that device and oracle agree bit for bit; against the
    exact power it is correctly rounded except for rare last-bit ties, i.e. it is the reference's std::pow for every purpose here."""
    rng = np.random.default_rng(1)
    n = 400000
    x = rng.random(n, dtype=np.float32)
    y = rng.choice(np.array([1, 2, 3, 5, 10, 50, 100, 500, 1000, 5000, 0.5, 0.25, 1 / 51.0, 1 / 501.0, 1.5, 2.5, 0.7], np.float32), n)
    tiny = np.finfo(np.float32).tiny

    def exact(a, b):
        with np.errstate(all="ignore"):
            r = np.power(a.astype(np.float64), b.astype(np.float64)).astype(np.float32)
        return np.where(np.abs(r) < tiny, np.float32(0) * r, r)   # FTZ like main.cpp:70-71

    def ulps(a, b):
        return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))

    u = ulps(oracle.powf(x, y), exact(x, y))
    assert u.max() <= 1 and (u > 0).mean() < 1e-4
    x = np.exp(rng.uniform(-40, 40, n)).astype(np.float32); y = rng.uniform(-20, 20, n).astype(np.float32)
    o, r = oracle.powf(x, y), exact(x, y)
    fin = np.isfinite(o) & np.isfinite(r)
    assert np.array_equal(np.isinf(o), np.isinf(r)) and ulps(o[fin], r[fin]).max() <= 1
    inf, nan = np.inf, np.nan
    xs = np.array([0, 0, 1, -1, -1, -2, -2, -2, inf, inf, 0.5, 2, 0.5, 2, nan, 3, -8, 1e-30, 1e30, -1, 7], np.float32)
    ys = np.array([2, -2, nan, 3, 2, 3, 2, 0.5, 2, -2, inf, inf, -inf, -inf, 0, nan, 1 / 3.0, 5, 5, inf, 0], np.float32)
    o, r = oracle.powf(xs, ys), exact(xs, ys)
    assert ((o == r) | (np.isnan(o) & np.isnan(r))).all(), (o, r)
    # frozen values: a change of the definition must show up here, not only as a device mismatch
    kx = np.array([0.5, 0.9, 0.999, 0.25, 0.75, 1e-3], np.float32); ky = np.array([50, 500, 5000, 1 / 51.0, 1.5, 0.7], np.float32)
    assert oracle.powf(kx, ky).view(np.uint32).tolist() == exact(kx, ky).view(np.uint32).tolist()


# ---- the reference's own inline tests beyond the cameras (tests/golden/reference_constants.json) ----
def _ulp_dist(a, b):
    """ulp_dist (unittest.cpp:150-166)."""
    ia, ib = int(np.float32(a).view(np.int32)), int(np.float32(b).view(np.int32))
    if (ia < 0) != (ib < 0):
        return abs(ia & 0x7FFFFFFF) + abs(ib & 0x7FFFFFFF)
    return abs(ia - ib)


def _almost_eq(a, b):
    """almost_eq (unittest.cpp:168-171)."""
    return bool(abs(np.float32(a) - np.float32(b)) < np.float32(1.1920928955078125e-07) or _ulp_dist(a, b) < 64)


def test_reference_almost_eq_vectors_and_float_environment():
    import json
    from conftest import ROOT
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_constants.json")))
    for v in g["almost_eq"]["vectors"]:  # unittest.cpp:177-182
        if "ulp_dist" in v:
            assert _ulp_dist(v["a"], v["b"]) == v["ulp_dist"]
        if v.get("ulp_dist_nonzero"):
            assert _ulp_dist(v["a"], v["b"]) != 0
        if "almost_eq" in v:
            assert _almost_eq(v["a"], v["b"]) == v["almost_eq"]
    # main.cpp:24-39 through the build's own transcendental definitions: sin(pi/2) = 1 (quadrant reduction of sincos_2pi at u = 1/4)
    env = g["float_environment"]
    orc = oracle.Oracle(load_scene("CornellBoxDiffuse"))
    assert _almost_eq(oracle.sincos_2pi(0.25)[0], env["sin_half_pi"]) and _almost_eq(oracle.sincos_2pi(0.25)[1], 0.0)
    assert _almost_eq(oracle.asinf(1.0), env["asin_1"])
    with np.errstate(divide="ignore", invalid="ignore"):
        assert np.isinf(np.float32(1) / np.float32(0)) and np.isnan(np.float32(0) / np.float32(0)) and np.isnan(np.float32(-0.0) / np.float32(0))
    # main.cpp:41-56 / SURVEY App. A: mat3 values cross the C ABI column-major — column 0 of the view-to-world frame is the camera's right vector
    f = ma.camera_setup(ma.Camera((ma.C.c_float * 3)(0, 0, 0), (ma.C.c_float * 3)(0, 0, -1), (ma.C.c_float * 3)(0, 1, 0), 1.0), 1.0)
    m = list(f.view_to_world)
    assert [_almost_eq(x, y) for x, y in zip(m, [1, 0, 0, 0, 1, 0, 0, 0, 1])] == [True] * 9
    assert g["test_scene_mean_radiance"]["status"].endswith("unpinned") and g["test_scene_mean_radiance"]["reference_script_constant"] == 0.01
    del orc


def test_converged_crop_fixtures_are_consistent():
    """The two 1024-spp oracle crops of BASELINE configs[1] (image-level parity rule of BASELINE.md) agree with each other within noise and
    with a fresh oracle render of the same window — the fixture is what today's oracle computes."""
    from conftest import ROOT
    a = np.load(os.path.join(ROOT, "tests", "golden", "c2_crop_1024spp_a.npy")); b = np.load(os.path.join(ROOT, "tests", "golden", "c2_crop_1024spp_b.npy"))
    assert a.shape == b.shape == (64, 64, 4) and np.all(a[..., 3] == 1024) and np.all(b[..., 3] == 1024)
    ra, rb = a[..., :3] / 1024, b[..., :3] / 1024
    assert abs(ra.mean() - rb.mean()) / ra.mean() < 0.005
    orc = oracle.Oracle(load_scene("CornellBoxDiffuse"), max_path=8)
    fresh = orc.render_rgbn(512, 512, spp=16, seed=101, window=(224, 160, 64, 64))[160:224, 224:288]
    first16 = orc.render_rgbn(512, 512, spp=1024, seed=101, window=(224, 160, 64, 64))[160:224, 224:288]
    assert np.array_equal(first16, a) and np.all(fresh[..., 3] == 16)
