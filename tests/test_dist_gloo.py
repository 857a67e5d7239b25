"""Multi-GPU path covered on CPU: world_size-2 `gloo` processes run master_amd.dist's sharding and
merge (the code bench.py runs over RCCL) with the oracle standing in for the device renderer.
Merged shards must equal one process rendering the union of the sample ranges (merge_exr semantics:
sum of R, G, B and denom, Options.cpp:1356-1358)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, scene_path


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, spp, steps, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import master_amd as ma
    import oracle
    from master_amd import dist as madist

    scene = ma.Scene.load(scene_path("CornellBoxDiffuse"))
    orc = oracle.Oracle(scene, max_path=4)
    total = torch.zeros((24, 24, 4), dtype=torch.float32)
    for step in range(steps):
        off = madist.sample_offset(step, rank, world, spp)
        fb = torch.from_numpy(orc.render_rgbn(24, 24, spp=spp, seed=5, sample_offset=off, threads=1))
        madist.merge_framebuffers(fb)          # all-reduce(sum)
        total += fb
    fb0 = torch.from_numpy(orc.render_rgbn(24, 24, spp=spp, seed=5, sample_offset=madist.sample_offset(0, rank, world, spp), threads=1))
    madist.merge_framebuffers(fb0, dst=0)      # reduce to rank 0
    np.save(os.path.join(out_dir, "total_%d.npy" % rank), total.numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, "reduce0.npy"), fb0.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _tile_worker(rank, world, port, spp, steps, out_dir):
    """Pixel-tile sharding (BASELINE C5): the rank renders ALL samples of its 32x32 tiles, zeros elsewhere."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import master_amd as ma
    import oracle
    from master_amd import dist as madist

    W, H = 72, 40
    scene = ma.Scene.load(scene_path("CornellBoxDiffuse"))
    orc = oracle.Oracle(scene, max_path=4)
    mine = madist.tile_owner(W, H, world) == rank
    total = torch.zeros((H, W, 4), dtype=torch.float32)
    for step in range(steps):
        off, n = madist.tile_sample_range(step, world, spp)
        full = orc.render_rgbn(W, H, spp=n, seed=5, sample_offset=off, threads=1)  # stand-in for the device's sharded render:
        fb = torch.from_numpy(np.where(mine[..., None], full, np.float32(0)))       # what mi_pt_set_tile_shard leaves in the framebuffer
        madist.merge_framebuffers(fb)
        total += fb
    np.save(os.path.join(out_dir, "tiles_%d.npy" % rank), total.numpy())
    dist.barrier()
    dist.destroy_process_group()


C5_W, C5_H = 3840, 2160                 # BASELINE configs[4]
C5_BAND = (0, 1024, 3840, 64)           # two rows of 32x32 tiles of that frame: 240 tiles, both ranks own 120


def _c5_worker(rank, world, port, out_dir):
    """The C5 shape through the N > 1 code path: clutter scene (BreakfastRoom1 stand-in), 3840x2160 frame, tile ownership over the
    8 160 tiles of the frame, one all-reduce of the full 126.6 MiB [H][W][4] framebuffer.  To stay within CPU-test time the
    oracle renders one band of the frame (the rest is zero on every rank, as for tiles a rank does not own)."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from master_amd import dist as madist
    from master_amd import scenegen

    orc = oracle.Oracle(scenegen.clutter())
    mine = madist.tile_owner(C5_W, C5_H, world) == rank
    off, n = madist.tile_sample_range(0, world, 1)
    full = orc.render_rgbn(C5_W, C5_H, spp=n, seed=5, sample_offset=off, window=C5_BAND, threads=2)
    fb = torch.from_numpy(np.where(mine[..., None], full, np.float32(0)))
    assert fb.numel() * 4 == 3840 * 2160 * 16
    madist.merge_framebuffers(fb)
    x0, y0, w, h = C5_BAND
    np.save(os.path.join(out_dir, "c5_%d.npy" % rank), fb.numpy()[y0:y0 + h])
    np.save(os.path.join(out_dir, "c5_outside_%d.npy" % rank), np.array([float(fb[:y0].abs().sum() + fb[y0 + h:].abs().sum()), float(mine.sum())]))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_tile_sharding_at_the_c5_shape(tmp_path):
    world = 2
    mp.spawn(_c5_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import oracle
    from master_amd import scenegen

    single = oracle.Oracle(scenegen.clutter()).render_rgbn(C5_W, C5_H, spp=world, seed=5, sample_offset=0, window=C5_BAND, threads=4)
    x0, y0, w, h = C5_BAND
    b0, b1 = np.load(tmp_path / "c5_0.npy"), np.load(tmp_path / "c5_1.npy")
    assert np.array_equal(b0, b1) and np.array_equal(b0, single[y0:y0 + h])  # one owner per pixel: the reduce only adds zeros
    o0, o1 = np.load(tmp_path / "c5_outside_0.npy"), np.load(tmp_path / "c5_outside_1.npy")
    assert o0[0] == 0 and o1[0] == 0 and o0[1] + o1[1] == C5_W * C5_H and abs(o0[1] - o1[1]) <= 32 * 32 * 120


def test_tile_owners_partition_the_window():
    from master_amd import dist as madist

    for world in (1, 2, 3, 8):
        for (W, H, win) in ((72, 40, None), (100, 70, (5, 3, 66, 64)), (31, 31, None)):
            own = madist.tile_owner(W, H, world, win)
            x0, y0, w, h = win if win else (0, 0, W, H)
            inside = np.zeros((H, W), bool); inside[y0:y0 + h, x0:x0 + w] = True
            assert ((own >= 0) == inside).all() and own.max() < world
            # tiles are 32x32 blocks counted from the window's origin, row-major, dealt round-robin
            assert own[y0, x0] == 0
            if w > 32:
                assert own[y0, x0 + 32] == 1 % world and own[y0 + min(31, h - 1), x0 + 31] == 0
            if h > 32:
                assert own[y0 + 32, x0] == ((w + 31) // 32) % world
    assert madist.tile_sample_range(2, 8, 16) == (256, 128)


def test_world2_gloo_tile_sharding_is_bit_identical_to_one_process(tmp_path):
    world, spp, steps = 2, 2, 2
    mp.spawn(_tile_worker, args=(world, _free_port(), spp, steps, str(tmp_path)), nprocs=world, join=True)
    import master_amd as ma
    import oracle

    scene = ma.Scene.load(scene_path("CornellBoxDiffuse"))
    orc = oracle.Oracle(scene, max_path=4)
    single = sum(orc.render_rgbn(72, 40, spp=spp * world, seed=5, sample_offset=s * spp * world, threads=1) for s in range(steps))
    t0, t1 = np.load(tmp_path / "tiles_0.npy"), np.load(tmp_path / "tiles_1.npy")
    assert np.array_equal(t0, t1) and np.array_equal(t0, single)  # every pixel has one owner: the reduce only adds zeros


def test_sample_offsets_partition_the_sample_axis():
    from master_amd import dist as madist

    for world in (1, 2, 4, 8):
        spp, steps = 16, 3
        seen = []
        for step in range(steps):
            for r in range(world):
                o = madist.sample_offset(step, r, world, spp)
                seen += list(range(o, o + spp))
        assert sorted(seen) == list(range(world * spp * steps))


def test_world2_gloo_merge_equals_single_process(tmp_path):
    world, spp, steps = 2, 3, 2
    mp.spawn(_worker, args=(world, _free_port(), spp, steps, str(tmp_path)), nprocs=world, join=True)
    import master_amd as ma
    import oracle

    scene = ma.Scene.load(scene_path("CornellBoxDiffuse"))
    single = oracle.Oracle(scene, max_path=4).render_rgbn(24, 24, spp=spp * world * steps, seed=5, sample_offset=0, threads=1)
    t0, t1 = np.load(tmp_path / "total_0.npy"), np.load(tmp_path / "total_1.npy")
    assert np.array_equal(t0, t1)                                  # all-reduce: every rank holds the merged image
    assert np.array_equal(t0[..., 3], single[..., 3])              # denom sums exactly
    np.testing.assert_allclose(t0, single, rtol=1e-6, atol=1e-7)   # FP32 partial sums vs one FP64 accumulation
    first = oracle.Oracle(scene, max_path=4).render_rgbn(24, 24, spp=spp * world, seed=5, sample_offset=0, threads=1)
    np.testing.assert_allclose(np.load(tmp_path / "reduce0.npy"), first, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("rank,local_rank,visible", [(1, 1, 2), (3, 3, 8), (1, 1, 1)])
def test_bench_rccl_branch_reaches_init_process_group(monkeypatch, rank, local_rank, visible):
    """the `--backend nccl` branch of bench.py with WORLD_SIZE > 1 has never run anywhere (one GPU per gpurun box).  Its
    plumbing up to the first collective is checked here without a GPU: RANK / LOCAL_RANK / WORLD_SIZE from the launcher's environment, LOCAL_RANK mapped
    onto the visible devices (one visible device per rank when the launcher restricts visibility), MASTER_ADDR defaulted to 127.0.0.1, and
    init_process_group called with backend "nccl" (= RCCL on ROCm), this rank, the world size and the rank's device."""
    import importlib
    world = 8 if rank == 3 else 2
    seen = {}

    class Reached(Exception):
        pass

    def fake_init(**kw):
        seen.update(kw)
        raise Reached()

    monkeypatch.setenv("RANK", str(rank)); monkeypatch.setenv("LOCAL_RANK", str(local_rank)); monkeypatch.setenv("WORLD_SIZE", str(world))
    monkeypatch.delenv("MASTER_ADDR", raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: visible)
    chosen = []
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: chosen.append(d))
    monkeypatch.setattr(dist, "init_process_group", fake_init)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", str(world), "--steps", "1", "--warmup", "0"])
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    with pytest.raises(Reached):
        bench.main()
    dev = local_rank % visible
    assert chosen == [dev]
    assert seen["backend"] == "nccl" and seen["rank"] == rank and seen["world_size"] == world and seen["device_id"] == torch.device("cuda", dev)
    assert os.environ["MASTER_ADDR"] == "127.0.0.1"
    # a launch whose WORLD_SIZE disagrees with --gpus is refused before anything is initialised
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit):
        bench.main()
