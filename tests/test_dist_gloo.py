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
