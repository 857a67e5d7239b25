#!/usr/bin/env python3
"""One-rank RCCL sanity on the GPU box: the collective bench.py uses (all-reduce of the [H][W][4] framebuffer) through
master_amd.dist with backend nccl.  A one-GPU box cannot run more ranks over RCCL; the N > 1 logic is covered by the gloo tests."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from master_amd import dist as madist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
fb = torch.full((512, 512, 4), 2.0, device="cuda")
dist.all_reduce(fb, op=dist.ReduceOp.SUM)   # the raw collective
madist.merge_framebuffers(fb)               # world 1: returns without a collective
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(); torch.cuda.synchronize()
assert float(fb.sum()) == 2.0 * 512 * 512 * 4 and float(t) == 1.5
print("rccl ok:", torch.cuda.nccl.version())
dist.destroy_process_group()
