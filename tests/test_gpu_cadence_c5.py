"""Round-2 parity cases (run with -m gpu on an MI355X):

* the frame cadence of the reference — Application::render calls Technique::render once per sample (Application.cpp:41-79,
  framework.cpp:426-437) — through mi_pt_render_async / mi_pt_wait: frames in flight give the same view, bit for bit, as the
  synchronous loop and as the oracle's frames;
* BASELINE configs[4] (C5) through its stand-in: the seeded `clutter` room at 3840x2160, pixel-tile shard of world size 8 —
  properties at full size plus windows inside one rank's tiles against the oracle.
"""
import os

import numpy as np
import pytest

import master_amd as ma
import oracle
from master_amd import dist as madist
from master_amd import scenegen as sb
from conftest import load_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("batch", [1, 4, 8])
@pytest.mark.parametrize("scene_name,kernel", [("CornellBoxDiffuse", ma.KERNEL_AUTO), ("CornellBoxSpecular", ma.KERNEL_AUTO), ("MetalRings", ma.KERNEL_AUTO),
                                               ("CornellBoxDiffuse", ma.KERNEL_WAVEFRONT)])
def test_frames_in_flight_give_the_synchronous_view_bit_for_bit(scene_name, kernel, batch):
    """render(k) = [enqueue a batch ahead] -> wait(frame k) -> view += rgbn: the dvec4 view after every call holds exactly frames 0..k
    (what --num-samples / snapshots / continue see), identical to one mi_pt_render(spp = 1) per frame — per frame and in total, ray
    counts included, whether the frames of a batch share one launch (frame variant of the megakernel) or not (wavefront pipeline)."""
    s = load_scene(scene_name)
    pt_a, pt_s = ma.PathTracing(s, max_path=6), ma.PathTracing(s, max_path=6)
    pt_a.set_kernel(kernel); pt_s.set_kernel(kernel)
    w, h, n = 96, 72, 11
    view_a, view_s = np.zeros((h, w, 4), np.float64), np.zeros((h, w, 4), np.float64)
    pt_a.render_frames(view_a, n, seed=9, batch=batch)
    for _ in range(n):
        pt_s.render(view_s, seed=9)
    assert np.array_equal(view_a, view_s) and np.all(view_a[..., 3] <= n)
    sa, ss = pt_a.statistics(), pt_s.statistics()
    assert (sa.num_samples, sa.num_basic_rays, sa.num_shadow_rays) == (ss.num_samples, ss.num_basic_rays, ss.num_shadow_rays)
    if scene_name == "CornellBoxDiffuse" and batch == 4:
        orc = oracle.Oracle(s, max_path=6)
        ref = np.zeros((h, w, 4), np.float64)
        for k in range(n):  # the oracle's frames, added one by one like Technique::_commit_images
            ref += orc.render_rgbn(w, h, spp=1, seed=9, sample_offset=k)
        np.testing.assert_allclose(view_a, ref, rtol=1.2e-7)


@pytest.mark.parametrize("frame_chunk,frame_tiles", [(None, None), ("1", "1"), ("3", "2"), ("8", "5")])
def test_batched_frames_equal_single_frames_with_exact_per_frame_statistics(monkeypatch, cornell, frame_chunk, frame_tiles):
    """mi_pt_render_frames_async: every frame of the launch is bit-identical to mi_pt_render(spp = 1) of that sample, and carries that
    frame's own ray / path / error counts (window and tile shard included), however frames and tiles are dealt to the waves."""
    if frame_chunk:
        monkeypatch.setenv("MI_PT_FRAME_CHUNK", frame_chunk); monkeypatch.setenv("MI_PT_FRAME_TILES", frame_tiles)
    spec = load_scene("CornellBoxSpecular")
    for scene, win, shard in ((cornell, None, None), (spec, (8, 4, 33, 21), None), (cornell, None, (1, 3))):
        pt = ma.PathTracing(scene, max_path=5)
        if shard:
            pt.set_tile_shard(*shard)
        tickets = pt.render_frames_async(64, 48, 7, seed=3, first_sample=10, window=win)
        assert tickets == list(range(tickets[0], tickets[0] + 7)) and pt.last_launch().frames == 7
        for f in (3, 0, 6, 1, 2, 5, 4):  # any order
            img = pt.wait(tickets[f])
            st = pt.last_stats
            ref = pt.render_rgbn(64, 48, spp=1, seed=3, sample_offset=10 + f, window=win)
            rs = pt.last_stats
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
            assert (st.num_paths, st.num_basic_rays, st.num_shadow_rays, st.numeric_errors) == (rs.num_paths, rs.num_basic_rays, rs.num_shadow_rays, rs.numeric_errors)


def test_async_tickets_slots_and_errors(cornell):
    pt = ma.PathTracing(cornell, max_path=4)
    t = [pt.render_async(40, 24, 1, 3, k) for k in range(ma.BATCHES_IN_FLIGHT)]
    assert t == list(range(t[0], t[0] + ma.BATCHES_IN_FLIGHT))
    with pytest.raises(ma.MiError) as e:  # every slot is pending
        pt.render_async(40, 24, 1, 3, 99)
    assert e.value.code == -1 and "pending" in str(e.value)
    frames = [pt.wait(k) for k in t]
    with pytest.raises(ma.MiError):  # already handed out
        pt.wait(t[0])
    with pytest.raises(ma.MiError):
        pt.render_frames_async(40, 24, ma.MAX_FRAMES_PER_BATCH + 1)
    for k, f in enumerate(frames):
        assert np.array_equal(f, pt.render_rgbn(40, 24, spp=1, seed=3, sample_offset=k))
    # a window, several samples per frame, statistics of the frame
    tk = pt.render_async(64, 48, 5, 3, 7, window=(8, 4, 33, 21))
    f = pt.wait(tk)
    st = pt.last_stats
    ref = pt.render_rgbn(64, 48, spp=5, seed=3, sample_offset=7, window=(8, 4, 33, 21))
    assert np.array_equal(f, ref) and st.num_paths == 33 * 21 * 5 == pt.last_stats.num_paths
    assert (st.num_basic_rays, st.num_shadow_rays) == (pt.last_stats.num_basic_rays, pt.last_stats.num_shadow_rays) and st.gpu_ms > 0


def test_last_launch_describes_the_launch(cornell):
    pt = ma.PathTracing(cornell, max_path=8)
    pt.render_rgbn(512, 512, spp=64, seed=1)
    li = pt.last_launch()
    assert li.kernel == ma.KERNEL_MEGA_LDS and li.n_blocks * 4 >= 64 * 64 * li.n_chunks and li.n_chunks * li.chunk_spp >= 64
    assert li.partial_bytes == li.n_chunks * 512 * 512 * 32 and li.lds_bytes <= 48 * 1024 and li.scene_bytes > 0
    big = ma.PathTracing(load_scene("LivingRoomLit"))
    big.render_rgbn(64, 64, spp=2, seed=1)
    lb = big.last_launch()
    assert lb.kernel == ma.KERNEL_MEGA_GLOBAL and lb.scene_bytes > 40000 * 15 * 16


# ---- BASELINE configs[4] (C5): BreakfastRoom1 stand-in, 3840x2160, pixel tiles sharded over 8 GPUs ----
C5_W, C5_H, C5_WORLD = 3840, 2160, 8


@pytest.fixture(scope="module")
def clutter():
    return sb.clutter()


@pytest.mark.parametrize("rank", [0, 5])
def test_c5_workload_tile_shard_properties_and_windows_against_oracle(clutter, rank):
    """One rank's share of C5': the 32x32 tiles {t : t mod 8 == rank} of the 3840x2160 frame, all samples of them (a bounded
    number here), zeros elsewhere.  Properties at full size: denom == spp exactly on the owned tiles and 0 elsewhere, path count
    = owned pixels x spp, no numeric errors lost, additivity over sample ranges; windows inside owned tiles equal the oracle's
    render of the same window of the same frame (pixel index, camera ray and stream all depend on the full resolution)."""
    spp = 4
    pt = ma.PathTracing(clutter)  # unbounded paths (reference default), Phong + mirror + glass, two area lights
    pt.set_tile_shard(rank, C5_WORLD)
    img = pt.render_rgbn(C5_W, C5_H, spp=spp, seed=0x5EED)
    st = pt.last_stats
    owner = madist.tile_owner(C5_W, C5_H, C5_WORLD)
    mine = owner == rank
    assert mine.sum() * C5_WORLD == pytest.approx(C5_W * C5_H, rel=0.02)
    assert np.all(img[~mine] == 0) and np.isfinite(img).all() and np.all(img[mine][:, 3] <= spp)
    assert np.mean(img[mine][:, 3] == spp) > 0.999  # glass without a TIR guard drops a few samples (BSDF.cpp:480-493, Technique.cpp:224)
    assert st.num_paths == int(mine.sum()) * spp and int(mine.sum()) * spp - int(img[..., 3].sum()) == st.numeric_errors
    assert pt.get_kernel() == ma.KERNEL_MEGA_GLOBAL and pt.last_launch().wide_nodes == 1  # >= 100 000 triangles: wide quantised nodes
    # additivity over sample ranges on the sharded render (what the framebuffer reduce relies on)
    p1, p3 = pt.render_rgbn(C5_W, C5_H, spp=1, seed=0x5EED, sample_offset=0).astype(np.float64), pt.render_rgbn(C5_W, C5_H, spp=3, seed=0x5EED, sample_offset=1)
    parts = p1 + p3
    assert np.array_equal(parts[..., 3], img[..., 3])
    # each framebuffer is one FP32 cast of an FP64 sum; a mirror seen from behind its shading normal contributes with a negative
    # sign (ReflectionBSDF: throughput 1 / omega.y, BSDF.cpp:450-465), so the bound is relative to the parts, not to their sum
    assert np.all(np.abs(parts - img) <= 1.2e-7 * (np.abs(p1) + np.abs(p3) + np.abs(img)))
    # windows of the frame inside this rank's tiles, against the oracle (per pixel: same paths, FP64 sum order aside)
    orc = oracle.Oracle(clutter)
    tiles_x = (C5_W + 31) // 32
    for t in (rank, rank + C5_WORLD * 531, rank + C5_WORLD * 1001):
        ty, tx = divmod(t, tiles_x)
        win = (tx * 32 + 3, ty * 32 + 5, 24, 20)
        x0, y0, w, h = win
        assert mine[y0:y0 + h, x0:x0 + w].all()
        ref = orc.render_rgbn(C5_W, C5_H, spp=spp, seed=0x5EED, window=win)
        np.testing.assert_allclose(img[y0:y0 + h, x0:x0 + w], ref[y0:y0 + h, x0:x0 + w], rtol=1.2e-7)


def test_c5_one_rank_renders_its_whole_share(clutter):
    """BASELINE configs[4] at the size ONE of the eight GPUs really gets: the 32x32 tiles {t : t mod 8 == 3} of the
    3840x2160 frame x ALL 4096 samples of them = 1/8 of the pixels x 4096 spp = the 512-spp-per-GPU weak-scaling unit of SURVEY 8(e),
    4.25 G paths in one call (Technique.cpp:167 tiles; the merge is the sum of Options.cpp:1340-1409).  Size-independent properties:
    denominators, exact path count, numeric errors == missing denominators, the two 2048-sample halves sum to it; plus one window of an
    owned tile against the oracle at a bounded sample count (pixel index, camera ray and streams depend on the full resolution)."""
    rank, spp = 3, 4096
    pt = ma.PathTracing(clutter)
    pt.set_tile_shard(rank, C5_WORLD)
    img = pt.render_rgbn(C5_W, C5_H, spp=spp, seed=0x5EED)
    st = pt.last_stats
    li = pt.last_launch()
    mine = madist.tile_owner(C5_W, C5_H, C5_WORLD) == rank
    n_mine = int(mine.sum())
    assert n_mine == C5_W * C5_H // C5_WORLD == 1005 * 1024 + 15 * 512  # 1 020 of the 8 160 tiles; 15 of them in the top row, which is 16 pixels high
    assert np.all(img[~mine] == 0) and np.isfinite(img).all()
    assert st.num_paths == n_mine * spp == 4246732800
    den = img[mine][:, 3].astype(np.float64)
    assert np.all(den <= spp) and np.mean(den == spp) > 0.5 and den.min() >= spp // 2 and den.sum() > 0.999 * n_mine * spp  # glass without a TIR guard drops samples (BSDF.cpp:480-493)
    assert n_mine * spp - int(den.sum()) == st.numeric_errors  # every dropped sample is a missing denominator (Technique.cpp:222-230)
    assert st.num_basic_rays > 3 * st.num_paths and st.num_shadow_rays > st.num_paths
    assert pt.get_kernel() == ma.KERNEL_MEGA_GLOBAL and li.wide_nodes == 1 and li.n_chunks * li.chunk_spp >= spp
    # the two halves of the sample range sum to the whole (what a resumed or merged render relies on); each framebuffer is one FP32 cast of an FP64 sum
    lo = pt.render_rgbn(C5_W, C5_H, spp=spp // 2, seed=0x5EED, sample_offset=0).astype(np.float64)
    paths_lo, err_lo, rays_lo = pt.last_stats.num_paths, pt.last_stats.numeric_errors, pt.last_stats.num_basic_rays
    hi = pt.render_rgbn(C5_W, C5_H, spp=spp // 2, seed=0x5EED, sample_offset=spp // 2).astype(np.float64)
    assert paths_lo + pt.last_stats.num_paths == st.num_paths and err_lo + pt.last_stats.numeric_errors == st.numeric_errors
    assert rays_lo + pt.last_stats.num_basic_rays == st.num_basic_rays
    parts = lo + hi
    assert np.array_equal(parts[..., 3], img[..., 3])
    assert np.all(np.abs(parts - img) <= 1.2e-7 * (np.abs(lo) + np.abs(hi) + np.abs(img)))
    # one window of an owned tile against the oracle, 8 samples of the same streams
    tiles_x = (C5_W + 31) // 32
    ty, tx = divmod(rank + C5_WORLD * 700, tiles_x)
    x0, y0, w, h = tx * 32 + 4, ty * 32 + 6, 24, 20
    assert mine[y0:y0 + h, x0:x0 + w].all()
    got = pt.render_rgbn(C5_W, C5_H, spp=8, seed=0x5EED)  # the sharded frame again (a window would renumber the tiles from its own origin)
    ref = oracle.Oracle(clutter).render_rgbn(C5_W, C5_H, spp=8, seed=0x5EED, window=(x0, y0, w, h))
    np.testing.assert_allclose(got[y0:y0 + h, x0:x0 + w], ref[y0:y0 + h, x0:x0 + w], rtol=1.2e-7)
    # the mean of the 4096-sample image over that window agrees with the 8-sample oracle render within its noise (same estimator, more samples)
    a = img[y0:y0 + h, x0:x0 + w]
    m4096, m8 = (a[..., :3] / a[..., 3:]).mean(), (ref[y0:y0 + h, x0:x0 + w, :3] / ref[y0:y0 + h, x0:x0 + w, 3:]).mean()
    assert abs(m4096 - m8) < 0.5 * max(m4096, m8)


def test_c5_tile_shards_partition_the_frame_bitwise(clutter):
    """The eight ranks' framebuffers of C5' at full resolution (1 spp) sum to the unsharded render bit for bit: every pixel has one owner."""
    pt = ma.PathTracing(clutter)
    whole = pt.render_rgbn(C5_W, C5_H, spp=1, seed=2)
    total = np.zeros_like(whole)
    paths = 0
    for r in range(C5_WORLD):
        pt.set_tile_shard(r, C5_WORLD)
        part = pt.render_rgbn(C5_W, C5_H, spp=1, seed=2)
        paths += pt.last_stats.num_paths
        assert np.all(total[part[..., 3] > 0] == 0)  # disjoint owners
        total += part
    assert paths == C5_W * C5_H and np.array_equal(total.view(np.uint32), whole.view(np.uint32))


@pytest.mark.parametrize("scene_name", ["CornellBoxDiffuse", "CornellBoxSpecular", "MetalRings"])
def test_frame_mode_writes_the_same_frame_as_the_accumulating_kernel(monkeypatch, scene_name):
    """spp == 1 runs the frame variant of the megakernel (paths write (r, g, b, 1) straight into the framebuffer, a wave regenerates over
    several 8x8 tiles); MI_PT_FRAME_MODE=0 forces the accumulating variant (LDS FP64 sums -> partial -> pt_finalize).  One sample per
    pixel: both must give the same bits, whatever the window, the tile shard and the tiles a wave owns."""
    s = load_scene(scene_name)
    pt = ma.PathTracing(s, max_path=7)
    cases = [(64, 64, None, None), (37, 23, None, None), (130, 41, (120, 3, 10, 30), None), (100, 72, None, (1, 3)), (96, 64, (7, 9, 70, 40), (0, 2)), (1, 1, None, None)]
    for (w, h, win, shard) in cases:
        pt.set_tile_shard(*(shard if shard else (0, 1)))
        monkeypatch.setenv("MI_PT_FRAME_MODE", "0")
        ref = pt.render_rgbn(w, h, spp=1, seed=4, sample_offset=5, window=win)
        ref_st = (pt.last_stats.num_paths, pt.last_stats.num_basic_rays, pt.last_stats.num_shadow_rays, pt.last_stats.numeric_errors)
        assert pt.last_launch().frame_tiles_per_wave == 0
        monkeypatch.delenv("MI_PT_FRAME_MODE")
        for tiles in ("1", "3", "4", "64"):
            monkeypatch.setenv("MI_PT_FRAME_TILES", tiles)
            img = pt.render_rgbn(w, h, spp=1, seed=4, sample_offset=5, window=win)
            assert pt.last_launch().frame_tiles_per_wave == int(tiles)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (w, h, win, shard, tiles)
            assert ref_st == (pt.last_stats.num_paths, pt.last_stats.num_basic_rays, pt.last_stats.num_shadow_rays, pt.last_stats.numeric_errors)
        monkeypatch.delenv("MI_PT_FRAME_TILES")


# ---- unified traversal with dynamic fetch (traverse_dyn): the default for scenes read from HBM ----
@pytest.mark.parametrize("name", ["LivingRoomLit", "MetalRings", "CornellBoxSpecular", "MirrorBalls", "soup20000", "CornellBoxDiffuse", "TestCaseFurnace", "atrium:120000"])
def test_dynamic_fetch_traversal_is_bit_identical_per_path(monkeypatch, name):
    """One loop over the closest-hit rays of trip k + 1 and the shadow rays of trip k, idle lanes fetching unstarted shadow rays of the wave: who
    walks a ray cannot matter — per-path radiance and ray counts equal the plain kernel's and the oracle's bit for bit; images differ by the
    order of a pixel's FP64 sum at most.  Both node formats of HBM-resident scenes, the wide walk (>= 100 000 triangles) and the LDS-resident
    variant (off by default there) are covered."""
    if name.startswith("soup"):
        s = sb.random_soup(int(name[4:]), seed=11)
    elif name.startswith("atrium"):
        s = sb.load(name)
    else:
        s = load_scene(name)
    pt = ma.PathTracing(s, max_path=9)
    w, h, spp = 48, 40, 6
    xy = np.stack(np.meshgrid(np.arange(w), np.arange(h)), -1).reshape(-1, 2).astype(np.uint32)
    xy, si = np.tile(xy, (spp, 1)), np.repeat(np.arange(spp, dtype=np.uint64), w * h)
    out = {}
    for dyn in ("0", "1"):
        monkeypatch.setenv("MI_PT_DYN", dyn)
        rad, cnt = pt.trace_paths(w, h, xy, si, seed=21)
        img = pt.render_rgbn(w, h, spp=spp, seed=21)
        li = pt.last_launch()
        frame = pt.render_rgbn(w, h, spp=1, seed=21, sample_offset=3)
        out[dyn] = (rad, cnt, img, frame, (pt.last_stats.num_basic_rays, pt.last_stats.num_shadow_rays), li.dynamic_fetch)
    a, b = out["0"], out["1"]
    assert a[5] == 0 and b[5] == 1
    same = np.isclose(a[0], b[0], rtol=0, atol=0, equal_nan=True)
    assert same.all() and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2][..., 3], b[2][..., 3]) and np.allclose(a[2], b[2], rtol=1.2e-7, atol=0, equal_nan=True)
    assert np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32)) and a[4] == b[4]  # one sample per pixel: nothing to reorder
    orad, ocnt = oracle.Oracle(s, max_path=9).trace_paths(w, h, xy, si, seed=21)
    assert np.isclose(b[0], orad, rtol=0, atol=0, equal_nan=True).all() and np.array_equal(b[1], ocnt)
    monkeypatch.delenv("MI_PT_DYN")
    pt.render_rgbn(w, h, spp=2, seed=1)
    assert pt.last_launch().dynamic_fetch == (0 if pt.get_kernel() == ma.KERNEL_MEGA_LDS else 1)


# ---- flat leaf list (traverse_flat): the default for LDS-resident scenes of <= 24 leaf links (r03) ----
@pytest.mark.parametrize("name", ["CornellBoxDiffuse", "CornellBoxPhong", "TestCaseFurnace", "TestCase1", "TestCase14", "DoubleLight", "MirrorAndAreaLight", "LightOverBox",
                                  "IndirectCubeNone", "soup7", "soup31"])
@pytest.mark.parametrize("beta", [1.0, 1.5])
def test_flat_leaf_list_is_bit_identical_per_path(monkeypatch, name, beta):
    """A uniform loop over the boxes of all leaf links (scalar operands), then per-lane tests of the leaves entered — no tree walk.  What is tested cannot
    change the (t, id) minimum or an occlusion: per-path radiance and ray counts equal the tree walk's and the oracle's bit for bit, images differ by the
    order of a pixel's FP64 sum at most; the one-sample frame mode is identical.  Covers single and pair leaves, light-only leaves (cut from a shadow
    ray's mask), Phong / mirror materials, several lights, a beta outside {1, 2} (general variant) and tables of 25..32 leaves (forced on)."""
    s = sb.random_soup(int(name[4:]), seed=5) if name.startswith("soup") else load_scene(name)
    pt = ma.PathTracing(s, max_path=9, beta=beta)
    w, h, spp = 48, 40, 6
    xy = np.stack(np.meshgrid(np.arange(w), np.arange(h)), -1).reshape(-1, 2).astype(np.uint32)
    xy, si = np.tile(xy, (spp, 1)), np.repeat(np.arange(spp, dtype=np.uint64), w * h)
    out = {}
    for flat in ("0", "1"):
        monkeypatch.setenv("MI_PT_FLAT", flat)
        rad, cnt = pt.trace_paths(w, h, xy, si, seed=21)
        img = pt.render_rgbn(w, h, spp=spp, seed=21)
        li = pt.last_launch()
        frame = pt.render_rgbn(w, h, spp=1, seed=21, sample_offset=3)
        pt.set_instrumented(True)
        ins = pt.render_rgbn(w, h, spp=spp, seed=21)
        ist = pt.last_stats
        pt.set_instrumented(False)
        out[flat] = (rad, cnt, img, frame, (pt.last_stats.num_basic_rays, pt.last_stats.num_shadow_rays), li.flat_leaves, ins, ist)
    a, b = out["0"], out["1"]
    assert a[5] == 0 and 1 <= b[5] <= 32
    assert np.isclose(a[0], b[0], rtol=0, atol=0, equal_nan=True).all() and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2][..., 3], b[2][..., 3]) and np.allclose(a[2], b[2], rtol=1.2e-7, atol=0, equal_nan=True)
    assert np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32)) and a[4] == b[4]
    # the instrumented variant renders the same image; it counts every box of the padded table per ray and two triangles per leaf entered
    assert np.array_equal(b[6][..., 3], b[2][..., 3]) and np.allclose(b[6], b[2], rtol=1.2e-7, atol=0, equal_nan=True)
    assert b[7].nodes_closest == b[7].num_basic_rays * ((b[5] + 3) // 4 * 4) and b[7].tris_closest % 2 == 0
    orad, ocnt = oracle.Oracle(s, max_path=9, beta=beta).trace_paths(w, h, xy, si, seed=21)
    assert np.isclose(b[0], orad, rtol=0, atol=0, equal_nan=True).all() and np.array_equal(b[1], ocnt)
    monkeypatch.delenv("MI_PT_FLAT")
    pt.render_rgbn(w, h, spp=2, seed=1)
    assert (pt.last_launch().flat_leaves != 0) == (8 <= b[5] <= 24)


@pytest.mark.parametrize("name", ["CornellBoxDiffuse", "CornellBoxPhong", "TestCaseFurnace", "TestCase0", "DoubleLight", "TestCase14", "LightOverBox", "soup31"])
def test_flat_leaf_list_hooks_on_adversarial_rays(monkeypatch, name):
    """Scene::intersect / occluded through the flat leaf list (MI_PT_INTERSECT_FLAT=1: the staging and traverse_flat of the megakernel) against the
    oracle's brute force and its tree: random rays, axis-parallel rays (1 / 0 replaced by a finite stand-in), rays inside the planes of the scene's
    quads (zero-thickness boxes), rays that start on vertices, on box corners and on the bounds of the table's padding.  The padded centre / half-extent
    boxes must not lose a hit: t, primitive and SurfacePoint bytes are identical (up to the grazing-ray ambiguity stated below)."""
    s = sb.random_soup(int(name[4:]), seed=5) if name.startswith("soup") else load_scene(name)
    monkeypatch.setenv("MI_PT_INTERSECT_FLAT", "1")
    pt, orc = ma.PathTracing(s), oracle.Oracle(s)
    monkeypatch.setenv("MI_PT_FLAT", "1")
    pt.render_rgbn(8, 8, spp=1, seed=1)
    assert pt.last_launch().flat_leaves > 0
    rng = np.random.default_rng(9)
    n = 60000
    lo, hi = s.positions.min(0), s.positions.max(0)
    o = np.zeros(n, ma.SURFACE_DTYPE)
    o["position"] = rng.uniform(lo, hi, (n, 3)); g = rng.normal(size=(n, 3)); o["gnormal"] = g / np.linalg.norm(g, axis=1, keepdims=True)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True); d = d.astype(np.float32)
    k = n // 6
    d[:k, rng.integers(0, 3, k)] = 0.0                                   # one zero component
    ax = rng.integers(0, 3, k); d[k:2 * k] = 0.0; d[np.arange(k, 2 * k), ax] = rng.choice([-1.0, 1.0], k)   # axis-parallel
    tri = s.indices[rng.integers(0, len(s.indices), k)]                  # start on vertices, head for other vertices (box corners, quad planes)
    o["position"][2 * k:3 * k] = s.positions[tri[:, 0]]
    tgt = s.positions[s.indices[rng.integers(0, len(s.indices), k), rng.integers(0, 3, k)]]
    dv = tgt - s.positions[tri[:, 0]]; nz = np.linalg.norm(dv, axis=1) > 0
    d[2 * k:3 * k][nz] = (dv[nz] / np.linalg.norm(dv[nz], axis=1, keepdims=True)).astype(np.float32)
    w = rng.uniform(0, 1, (k, 3)); w /= w.sum(1, keepdims=True)          # start inside triangles, leave within their planes
    tri2 = s.indices[rng.integers(0, len(s.indices), k)]
    p = (s.positions[tri2] * w[:, :, None]).sum(1)
    o["position"][3 * k:4 * k] = p
    e = s.positions[tri2[:, 1]] - s.positions[tri2[:, 0]]; en = np.linalg.norm(e, axis=1, keepdims=True); en[en == 0] = 1
    d[3 * k:4 * k] = (e / en).astype(np.float32)
    o["position"][4 * k:5 * k] = np.where(rng.integers(0, 2, (k, 3)) == 0, lo, hi)   # corners of the scene box

    def tg_of(o):
        tg = np.zeros(n, ma.SURFACE_DTYPE)
        tg["position"] = np.roll(o["position"], 17, axis=0); tg["gnormal"] = np.roll(o["gnormal"], 5, axis=0)
        return tg

    # A ray that grazes a triangle (|den| -> 0) can be accepted by the FP32 Moeller-Trumbore test at a point a few 1e-6 OUTSIDE the triangle's box — no
    # finite padding makes a box test agree with brute force there, and the walks (oracle tree, device tree with pair leaves, flat list) may then open
    # different boxes.  So: the flat list equals the oracle's brute force or its tree on EVERY ray, both on all but a handful of these edge-on rays, and
    # whenever tree and brute force agree it agrees with them (random path rays never get there: 0 mismatches in 17.7 M paths, profiles/r02/parity_sweep.txt).
    gh, gt, gp = pt.intersect(o, d)
    oh, ot, op = orc.intersect(o, d)
    gv, ov = pt.occluded(o, tg_of(o)), orc.occluded(o, tg_of(o))
    orc.set_use_bvh(False)                                               # brute force: no box test at all
    bh, bt, bp = orc.intersect(o, d)
    bv = orc.occluded(o, tg_of(o))
    tree_ok = (gp == op) & (gt == ot); brute_ok = (gp == bp) & (gt == bt)
    assert (tree_ok | brute_ok).all() and (tree_ok & brute_ok).sum() >= n - 12
    assert all(gh[i].tobytes() == (oh[i] if tree_ok[i] else bh[i]).tobytes() for i in np.nonzero(~(tree_ok & brute_ok))[0])
    both = tree_ok & brute_ok
    assert gh[both].tobytes() == oh[both].tobytes()
    assert ((gv == ov) | (gv == bv)).all() and ((gv == ov) & (gv == bv)).sum() >= n - 12
    monkeypatch.setenv("MI_PT_INTERSECT_FLAT", "0")                      # the device's tree walk obeys the same statement
    th, tt, tp = pt.intersect(o, d)
    assert (((tp == op) & (tt == ot)) | ((tp == bp) & (tt == bt))).all() and ((tp == gp) & (tt == gt)).sum() >= n - 12


def test_bench_two_ranks_share_the_gpu_on_the_c5_shape(tmp_path):
    """the N > 1 path of bench.py rehearsed on this box — two ranks (gloo, both on the one GPU), the C5 stand-in at 3840x2160 with the
    pixel-tile shard: launcher contract (torch.distributed.run, 127.0.0.1), barrier + max-over-ranks timing, one JSON line on rank 0 that carries the
    per-rank render / reduce times and the size of the reduce (126.6 MiB)."""
    import json
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--scene", "clutter", "--width", "3840", "--height", "2160", "--spp", "2",
           "--max-path", "0", "--shard", "tiles", "--backend", "gloo", "--share-gpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["unit"] == "Msamples/s"
    pr = d["per_rank"]
    assert len(pr["render_ms"]) == len(pr["reduce_ms"]) == len(pr["kernel_ms"]) == 2 and pr["reduce_bytes"] == 3840 * 2160 * 16
    assert all(t > 0 for t in pr["render_ms"] + pr["reduce_ms"]) and "32x32 pixel tiles" in d["config"]["parallelism"]
    assert "hbm_workload" not in d and "cpu_baseline" not in d and d["roofline"]["frac"] <= 1.0


# ---- pair leaves (bvh_build.hip k_pair_*): a node over two triangles becomes one leaf link over re-ordered streams ----
@pytest.mark.parametrize("name", ["CornellBoxDiffuse", "TestCase8", "CornellBoxSpecular", "LivingRoomLit", "MetalRings", "soup3000", "atrium:120000"])
def test_pair_leaves_are_bit_identical_per_path(monkeypatch, name):
    """MI_PT_PAIRS=0 builds the round-1 tree.  With pair leaves the walks open fewer nodes and test both triangles of a pair — the hit is the (t, id)
    minimum and occlusion a boolean either way: per-path radiance, ray counts, frames and BPT paths are the same bits (LDS-resident walk, float / binary
    quantised / wide quantised records read from HBM, unified loop), and the tree a caller downloads is the builder's in both cases."""
    if name.startswith("soup"):
        s = sb.random_soup(int(name[4:]), seed=5)
    elif name.startswith("atrium"):
        s = sb.load(name)
    else:
        s = load_scene(name)
    w, h, spp = 40, 32, 5
    xy = np.stack(np.meshgrid(np.arange(w), np.arange(h)), -1).reshape(-1, 2).astype(np.uint32)
    xy, si = np.tile(xy, (spp, 1)), np.repeat(np.arange(spp, dtype=np.uint64), w * h)
    out = {}
    for pairs in ("0", "1"):
        monkeypatch.setenv("MI_PT_PAIRS", pairs)  # read when the handle is created
        pt = ma.PathTracing(s, max_path=9, beta=2.0)
        rad, cnt = pt.trace_paths(w, h, xy, si, seed=17)
        frame = pt.render_rgbn(w, h, spp=1, seed=17, sample_offset=2)
        bpt = pt.bpt_trace_paths(w, h, xy[:2000], si[:2000], seed=3)
        nodes, order, morton = pt.bvh()
        out[pairs] = (rad, cnt, frame, bpt, nodes, order, morton)
    a, b = out["0"], out["1"]
    assert np.isclose(a[0], b[0], rtol=0, atol=0, equal_nan=True).all() and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    assert np.array_equal(a[3][2], b[3][2])
    for k in (0, 1):
        assert np.array_equal(np.asarray(a[3][k]).view(np.uint32), np.asarray(b[3][k]).view(np.uint32))
    assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6])  # the downloaded tree does not change
    if len(a[4]) > 1:  # ... and there is something to pair: some node has two leaf children
        assert ((a[4]["link0"] < 0) & (a[4]["link1"] < 0)).any()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 9, 17])
def test_tiny_trees_with_pair_leaves_against_the_oracle(monkeypatch, n):
    """n surface triangles + the two of the light quad: trees of two to nineteen leaves, where almost every node's child is a node over two triangles (a root
    over two leaves is the one bottom node that has no parent to carry its pair link).  Closest hits, occlusion and per-path radiance against the oracle,
    through the LDS-resident walk and — forced — the records read from HBM (float, binary quantised, wide quantised)."""
    s = sb.random_soup(n, seed=100 + n, extent=1.5, size=0.8)
    rng = np.random.default_rng(n)
    o = rng.uniform(-2.5, 2.5, (20000, 3)).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    tg = rng.uniform(-2.5, 2.5, (20000, 3)).astype(np.float32)
    orc = oracle.Oracle(s, max_path=6)
    oh, ot, op = orc.intersect(o, d)
    oo = orc.occluded(o, tg)
    w, h, spp = 24, 16, 4
    xy = np.stack(np.meshgrid(np.arange(w), np.arange(h)), -1).reshape(-1, 2).astype(np.uint32)
    xy, si = np.tile(xy, (spp, 1)), np.repeat(np.arange(spp, dtype=np.uint64), w * h)
    orad, ocnt = orc.trace_paths(w, h, xy, si, seed=9)
    for kernel, env in ((ma.KERNEL_AUTO, {}), (ma.KERNEL_MEGA_GLOBAL, {}), (ma.KERNEL_MEGA_GLOBAL, {"MI_PT_WIDE_NODES": "1"}),
                        (ma.KERNEL_MEGA_GLOBAL, {"MI_PT_WIDE_NODES": "0", "MI_PT_FLOAT_NODES": "1"}), (ma.KERNEL_WAVEFRONT, {})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pt = ma.PathTracing(s, max_path=6)
        pt.set_kernel(kernel)
        gh, gt, gp = pt.intersect(o, d)
        assert np.array_equal(gp, op) and np.array_equal(gt, ot) and gh.tobytes() == oh.tobytes(), (kernel, env)
        assert np.array_equal(pt.occluded(o, tg), oo), (kernel, env)
        rad, cnt = pt.trace_paths(w, h, xy, si, seed=9)
        assert np.isclose(rad, orad, rtol=0, atol=0, equal_nan=True).all() and np.array_equal(cnt, ocnt), (kernel, env)
        for k in env:
            monkeypatch.delenv(k)


def test_rccl_reduce_through_the_c_abi_world_of_one(cornell):
    """the reduce of the per-GPU framebuffers as a call of the library (north_star: "RCCL reduce over xGMI"), for hosts that run one
    process per GPU — mi_pt_reduce_unique_id / _init / _rgbn / _finalize dlopen the librccl beside the HIP runtime in use and run ncclAllReduce /
    ncclReduce(sum, f32, H*W*4) in place on the device framebuffer of mi_pt_render_device (merge_exr, Options.cpp:1340-1409).  One GPU per box: the
    communicator has ONE rank here (initialising it and reducing over it is legal and runs the real RCCL code path); the sum over one rank is the
    identity.  UNMEASURED across GPUs.  The device buffers come from the HIP runtime the library itself runs on (ctypes), like a C host's would: a
    framework with its own bundled ROCm stack in the same process is exactly what the loader steps around (mi_pt_api.hip rccl())."""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")  # the already loaded runtime of libmi_pt.so
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]; hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipDeviceSynchronize.argtypes = []

    def dev_alloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), nbytes) == 0
        return p

    def download(p, shape):
        out = np.empty(shape, np.float32)
        assert hip.hipDeviceSynchronize() == 0 and hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, out.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return out

    assert ma.reduce_available()
    pt = ma.PathTracing(cornell, max_path=5)
    w, h = 96, 64
    fb = dev_alloc(w * h * 16)
    with pytest.raises(ma.MiError) as e:  # no communicator yet
        pt.reduce_rgbn(fb.value, w, h)
    assert e.value.code == -1
    uid = ma.reduce_unique_id()
    assert len(uid) == ma.REDUCE_ID_BYTES and any(uid)
    with pytest.raises(ma.MiError):
        pt.reduce_init(uid, 1, 1)  # rank must be < world
    pt.reduce_init(uid, 0, 1)
    with pytest.raises(ma.MiError):
        pt.reduce_init(uid, 0, 1)  # one communicator per handle
    pt.render_device(fb.value, w, h, spp=8, seed=3)
    before = download(fb, (h, w, 4))
    assert np.array_equal(before, pt.render_rgbn(w, h, spp=8, seed=3))
    pt.reduce_rgbn(fb.value, w, h, root=-1)   # all-reduce on the handle's stream (synchronised)
    pt.reduce_rgbn(fb.value, w, h, root=0)    # reduce to rank 0
    after = download(fb, (h, w, 4))
    assert np.array_equal(after, before) and np.all(after[..., 3] == 8)
    with pytest.raises(ma.MiError):
        pt.reduce_rgbn(fb.value, w, h, root=1)  # root outside the communicator
    # the C5 payload: 3840 x 2160 x 4 f32 = 126.6 MiB in one call
    big = dev_alloc(3840 * 2160 * 16)
    ones = np.ones((2160, 3840, 4), np.float32)
    assert hip.hipMemcpy(big, ones.ctypes.data_as(C.c_void_p), ones.nbytes, 1) == 0  # hipMemcpyHostToDevice
    pt.reduce_rgbn(big.value, 3840, 2160)
    assert np.array_equal(download(big, (2160, 3840, 4)), ones)
    pt.reduce_finalize()
    pt.reduce_finalize()  # idempotent
    with pytest.raises(ma.MiError):
        pt.reduce_rgbn(fb.value, w, h)
    hip.hipFree(fb); hip.hipFree(big)
