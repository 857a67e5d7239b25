"""Multi-GPU sharding of the PT hot path: one process per GPU, samples sharded, one sum-reduce.

The path shards with no data-path exchange: every (pixel, sample) path is independent and the
random stream is keyed on the GLOBAL sample index, so rank r of R renders the sample range
[(step * R + r) * spp, (step * R + r + 1) * spp) of every pixel and the per-rank framebuffers
([H][W][4] float32 = RGB sums + sample count) are summed — exactly what the reference's offline
`master merge` does with EXRs from different machines (merge_exr, Options.cpp:1340-1409:
dst = fst + snd on (R, G, B, denom)).  The sum is one RCCL all-reduce over xGMI
(torch.distributed backend "nccl" on ROCm) or gloo on CPU tensors in the tests.

The second decomposition (BASELINE config C5) shards PIXELS: the window is cut
into the 32x32 tiles of Technique::_trace_paths (Technique.cpp:167), rank r owns the tiles
{t : t mod R == r} (mi_pt_set_tile_shard) and renders ALL samples of them; the other pixels are
zeros, so the same sum-reduce acts as a gather and the merged image is bit-identical to the
one-GPU render (one FP64 accumulation per pixel, on its owner).
"""

TILE = 32  # exec2d tile size, Technique.cpp:167


def sample_offset(step, rank, world_size, spp_per_rank):
    """First global sample index rendered by `rank` in `step` (weak scaling: spp_per_rank fixed)."""
    return (step * world_size + rank) * spp_per_rank


def merge_framebuffers(fb, group=None, dst=None):
    """Sum-reduce the [H][W][4] framebuffer over ranks (merge_exr semantics).  In place.
    dst=None: all-reduce (every rank gets the merged image); dst=r: reduce to rank r."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return fb
    if dst is None:
        dist.all_reduce(fb, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return fb


def tile_owner(width, height, world_size, window=None):
    """[h][w] int array: the rank that owns each pixel of the window under pixel-tile sharding
    (tiles numbered row-major from the window's origin, row 0 = bottom like the framebuffer)."""
    import numpy as np

    x0, y0, w, h = window if window else (0, 0, width, height)
    ty, tx = np.meshgrid(np.arange(h) // TILE, np.arange(w) // TILE, indexing="ij")
    owner = np.full((height, width), -1, dtype=np.int64)
    owner[y0:y0 + h, x0:x0 + w] = (ty * ((w + TILE - 1) // TILE) + tx) % max(world_size, 1)
    return owner


def tile_sample_range(step, world_size, spp_per_rank):
    """(sample_offset, spp) every rank renders on ITS tiles in `step` under pixel-tile sharding: each rank has
    1/world of the pixels and renders world x spp_per_rank samples of them (weak scaling: work per rank fixed)."""
    n = world_size * spp_per_rank
    return step * n, n
