"""Multi-GPU sharding of the PT hot path: one process per GPU, samples sharded, one sum-reduce.

The path shards with no data-path exchange: every (pixel, sample) path is independent and the
random stream is keyed on the GLOBAL sample index, so rank r of R renders the sample range
[(step * R + r) * spp, (step * R + r + 1) * spp) of every pixel and the per-rank framebuffers
([H][W][4] float32 = RGB sums + sample count) are summed — exactly what the reference's offline
`master merge` does with EXRs from different machines (merge_exr, Options.cpp:1340-1409:
dst = fst + snd on (R, G, B, denom)).  The sum is one RCCL all-reduce over xGMI
(torch.distributed backend "nccl" on ROCm) or gloo on CPU tensors in the tests.
"""


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
