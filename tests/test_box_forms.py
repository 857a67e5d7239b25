"""CPU property tests of the centre / half-extent box tests the round-3 kernels run (pt_device.h: traverse_flat, ce_box_test, wide_child_test).
The kernels' results do not depend on which boxes a walk opens — as long as a box test NEVER rejects a box the ray really enters.  The padding that
replaces the old per-ray slack term is argued in DESIGN.md §5a / §5c; here it is checked numerically: the tests are restated in float32 (fma = one
rounding of the exact product-sum; 1 / d perturbed by an ulp like v_rcp_f32) and thrown at adversarial rays — through box corners, along edges and faces,
with zero and tiny direction components — against the exact (float64) slab test of the unpadded box.  No GPU, no oracle."""
import numpy as np

F = np.float32


def fma(a, b, c):
    """float32 fma: the product of two float32 is exact in float64; one rounding of the sum to float64, then to float32"""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def exact_enters(lo, hi, o, d, tmax):
    """float64 slab test of the exact box against the exact ray (what 'the ray enters the box within [0, tmax]' means)"""
    lo, hi, o, d = (x.astype(np.float64) for x in (lo, hi, o, d))
    tn = np.zeros(len(o)); tf = np.full(len(o), tmax, np.float64)
    ok = np.ones(len(o), bool)
    for a in range(3):
        par = d[:, a] == 0
        inside = (o[:, a] >= lo[:, a]) & (o[:, a] <= hi[:, a])
        ok &= ~par | inside
        with np.errstate(divide="ignore", invalid="ignore"):
            t0 = (lo[:, a] - o[:, a]) / d[:, a]; t1 = (hi[:, a] - o[:, a]) / d[:, a]
        t0, t1 = np.where(par, -np.inf, np.minimum(t0, t1)), np.where(par, np.inf, np.maximum(t0, t1))
        tn = np.maximum(tn, t0); tf = np.minimum(tf, t1)
    return ok & (tn <= tf)


def device_inv(d, rng):
    """make_raybox: v_rcp_f32 (1 ulp), +-1e18 for zero / tiny components"""
    with np.errstate(divide="ignore"):
        inv = (F(1) / d).astype(F)
    inv = np.nextafter(inv, np.where(rng.integers(0, 2, inv.shape) == 0, F(np.inf), F(-np.inf)).astype(F)).astype(F)  # an ulp either way
    big = F(1e18)
    return np.where(np.abs(inv) < big, inv, np.copysign(big, d)).astype(F)


def ce_test(c, e, o, inv, tmax, slack=None):
    """ce_box_test / traverse_flat's box loop / wide_child_test in float32"""
    oi = (o * inv).astype(F)
    tn = np.zeros(len(o), F); tf = np.full(len(o), tmax, F)
    for a in range(3):
        m = fma(c[:, a], inv[:, a], -oi[:, a])
        tn = np.maximum(tn, fma(-e[:, a], np.abs(inv[:, a]), m))
        tf = np.minimum(tf, fma(e[:, a], np.abs(inv[:, a]), m))
    if slack is not None:
        s = ((np.abs(oi[:, 0]) + np.abs(oi[:, 1])).astype(F) + np.abs(oi[:, 2])).astype(F) * F(2.5e-7)
        return tn <= fma(tf, np.full(len(o), 1.000002, F), s.astype(F))
    return tn <= tf


def adversarial_rays(lo, hi, amax, rng, origin_lo, origin_hi):
    """origins anywhere in the allowed range; targets ON the box: corners, edge points, face points, interior; some directions get zero / tiny components"""
    n = len(lo)
    o = rng.uniform(origin_lo, origin_hi, (n, 3)).astype(F)
    w = rng.uniform(0, 1, (n, 3))
    kind = rng.integers(0, 4, (n, 1))  # 0 corner, 1 edge, 2 face, 3 interior: snap that many coordinates to a bound
    snap = np.argsort(rng.uniform(size=(n, 3)), axis=1) < (3 - kind)
    w = np.where(snap, rng.integers(0, 2, (n, 3)), w)
    tgt = (lo.astype(np.float64) + w * (hi.astype(np.float64) - lo.astype(np.float64))).astype(F)
    d = (tgt - o).astype(F)
    z = rng.uniform(size=(n, 3)) < 0.08
    d = np.where(z, F(0), d).astype(F)
    t = rng.uniform(size=(n, 3)) < 0.04
    d = np.where(t, (d * F(1e-7)).astype(F), d).astype(F)
    on_box = rng.uniform(size=n) < 0.15  # rays that START on the box (nudged origins of path rays sit 1e-4 off a face)
    o = np.where(on_box[:, None], tgt, o).astype(F)
    d = np.where(on_box[:, None], rng.normal(size=(n, 3)), d).astype(F)
    return o, d


def test_padded_float_boxes_never_lose_a_box_the_ray_enters():
    """The flat leaf table (host: mi_pt_create) and the LDS node copy (stage_scene_to_lds): c = (lo + hi) / 2, e = max(hi - c, c - lo) (* 1.000001) +
    2^-20 * amax, rays starting anywhere within [-amax, amax]^3; tmax = infinity (closest-hit rays) and 1 (shadow rays)."""
    rng = np.random.default_rng(5)
    for amax in (4.1, 400.0, 0.01):
        n = 400000
        a = rng.uniform(-amax, amax, (n, 3)); b = a + rng.uniform(0, 1, (n, 3)) ** 3 * amax * rng.choice([0.0, 1e-4, 0.02, 0.5], (n, 3))  # flat, thin and fat boxes
        lo, hi = np.minimum(a, b).clip(-amax, amax).astype(F), np.maximum(a, b).clip(-amax, amax).astype(F)
        pad = F(amax * 2.0 ** -20 + 1e-30)
        c = ((lo + hi) * F(0.5)).astype(F)
        e = fma(np.maximum(hi - c, c - lo).astype(F), np.full_like(c, 1.000001), np.full_like(c, pad))
        o, d = adversarial_rays(lo, hi, amax, rng, -amax, amax)
        inv = device_inv(d, rng)
        for tmax in (np.inf, 1.0):
            must = exact_enters(lo, hi, o, d, tmax)
            got = ce_test(c, e, o, inv, F(tmax))
            nan_dir = ~np.isfinite(d).all(1)
            lost = must & ~got & ~nan_dir
            assert not lost.any(), (amax, tmax, int(lost.sum()), o[lost][:2], d[lost][:2], lo[lost][:2], hi[lost][:2])
            assert must.mean() > 0.3  # the rays do aim at their boxes
            if tmax == np.inf:  # the test has teeth: the same boxes WITHOUT the padding lose a quarter of these rays, a quarter of the padding still a few
                bare = ce_test(c, np.maximum(hi - c, c - lo).astype(F), o, inv, F(tmax))
                assert (must & ~bare & ~nan_dir).sum() > 1000


def test_centre_form_copy_of_the_float_nodes_keeps_its_slack():
    """sv.ce_nodes (k_ce_nodes): padding 2^-21 of the box's own coordinates + the per-ray slack of the test — origins may be far from the box."""
    rng = np.random.default_rng(6)
    n = 400000
    a = rng.uniform(-2, 2, (n, 3)) * rng.choice([1.0, 100.0], (n, 1)); b = a + rng.uniform(0, 1, (n, 3)) ** 3 * rng.choice([0.0, 1e-3, 0.1, 1.0], (n, 3))
    lo, hi = np.minimum(a, b).astype(F), np.maximum(a, b).astype(F)
    c = ((lo + hi) * F(0.5)).astype(F)
    p = (np.maximum(np.abs(lo), np.abs(hi)) * F(2.0 ** -21)).astype(F)
    e = fma(np.maximum(hi - c, c - lo).astype(F), np.full_like(c, 1.000001), p)
    o, d = adversarial_rays(lo, hi, 400.0, rng, -400.0, 400.0)
    inv = device_inv(d, rng)
    for tmax in (np.inf, 1.0):
        must = exact_enters(lo, hi, o, d, tmax)
        got = ce_test(c, e, o, inv, F(tmax), slack=True)
        lost = must & ~got & np.isfinite(d).all(1)
        assert not lost.any(), (tmax, int(lost.sum()), o[lost][:2], d[lost][:2], lo[lost][:2], hi[lost][:2])


def test_quantised_children_never_lose_a_box_the_ray_enters():
    """wide_child_test (k_collapse4 / k_quantize): centre = (ql + qh) >> 1 and half extent = what covers both ends + 1 cell, on the 65536^3 grid; the ray is
    in grid space, its origin inside the grid."""
    rng = np.random.default_rng(7)
    n = 400000
    ql = rng.integers(0, 65535, (n, 3)); qh = np.minimum(ql + rng.integers(0, 2, (n, 3)) * rng.integers(0, 4000, (n, 3)) + rng.integers(0, 3, (n, 3)), 65535)
    cq = (ql + qh) >> 1
    eq = np.maximum(qh - cq, cq - ql) + 1
    lo, hi, c, e = ql.astype(F), qh.astype(F), cq.astype(F), eq.astype(F)
    o, d = adversarial_rays(lo, hi, 65535.0, rng, 0.0, 65535.0)
    inv = device_inv(d, rng)
    for tmax in (np.inf, 1.0):
        must = exact_enters(lo, hi, o, d, tmax)
        got = ce_test(c, e, o, inv, F(tmax))
        lost = must & ~got & np.isfinite(d).all(1)
        assert not lost.any(), (tmax, int(lost.sum()), o[lost][:2], d[lost][:2], lo[lost][:2], hi[lost][:2])
    bare = ce_test(c, np.maximum(qh - cq, cq - ql).astype(F), o, inv, F(np.inf))  # without the extra cell: thousands of boxes lost
    assert (exact_enters(lo, hi, o, d, np.inf) & ~bare & np.isfinite(d).all(1)).sum() > 1000


def test_any_hit_on_a_unit_segment_needs_no_division():
    """Scene::occluded casts a SEGMENT: tnear 0, tfar 1 exactly (Scene.cpp:165-175), and rtcOccluded accepts a triangle with t = T / |den| <= tfar.
    For floats T, |den| > 0 the correctly rounded quotient is <= 1 exactly when T <= |den| (T > |den| puts the quotient at least one ulp-ratio above 1,
    beyond the rounding midpoint), so the any-hit tests of the device compare T with |den| and skip the IEEE division (pt_device.h tri_test / flat_tri /
    traverse_dyn, r04).  Checked here on every neighbour of |den| within 4 ulps over 29 decades, powers of two included."""
    rng = np.random.default_rng(1)
    with np.errstate(all="ignore"):
        for scale in (1e-30, 1e-10, 1e-3, 1.0, 1e3, 1e20):
            d = (rng.random(400_000).astype(np.float32) + np.float32(1e-3)) * np.float32(scale)
            d = np.concatenate([d, np.float32(scale) * np.float32(2.0) ** rng.integers(-3, 3, 500).astype(np.float32)])
            for k in range(-4, 5):
                t = (d.view(np.uint32).astype(np.int64) + k).astype(np.uint32).view(np.float32)
                assert np.array_equal(t / d <= np.float32(1.0), t <= d)
