"""Data-parallel plumbing for the training step (one process per GPU, torch.distributed over RCCL/xGMI).

Scenes are independent units: the forward needs no exchange and inference runs plain replicas
(SURVEY.md 8e).  The training step has exactly one real exchange: the sum of the gradients.  The
reference intended DDP(find_unused_parameters=True) but never wired it (train.py:156-185); here the
gradients of ALL parameters live in one flat fp32 buffer (8 120 459 floats = 32.5 MB) that is
all-reduced in a few large buckets -- parameters that received no gradient (decoder/head before
``prepare_epochs``) contribute zeros, so every rank issues the identical collectives.
xGMI is point-to-point (7 links x ~153 GB/s): 32.5 MB is ~0.4 ms as a ring, so buckets are kept large
(default 8 MB) and the first ones start while the backbone's backward is still running.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")  # "nccl" IS RCCL on ROCm
    dist.init_process_group(backend)
    return world


class FlatGradAllReduce:
    """Gradient averaging over one flat buffer, bucketed."""

    def __init__(self, module, bucket_bytes=8 << 20, only_trainable=True):
        self.params = [p for p in module.parameters() if (p.requires_grad or not only_trainable)]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.bucket = max(int(bucket_bytes) // 4, 1)

    def numel(self):
        return self.flat.numel()

    @torch.no_grad()
    def reduce(self):
        """Average .grad across ranks (missing grads count as zero) and write the result back."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is not None:
                v.copy_(p.grad)
        if world > 1:
            works = [dist.all_reduce(self.flat[s:s + self.bucket], async_op=True)
                     for s in range(0, self.flat.numel(), self.bucket)]
            for w in works:
                w.wait()
            self.flat.div_(world)
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)


def max_over_ranks(seconds: float, device=None) -> float:
    """Wall time of the slowest rank (bench.py's timing rule)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
