"""Data-parallel plumbing for the training step (one process per GPU, torch.distributed over RCCL/xGMI).

Scenes are independent units: the forward needs no exchange and inference runs plain replicas
(SURVEY.md 8e).  The training step has exactly one real exchange: the sum of the gradients.  The
reference intended SyncBatchNorm + DDP(find_unused_parameters=True) but never wired them (train.py:156-185).

``BucketedGradReducer``: the gradients of all trainable parameters are exchanged through ONE flat fp32 buffer
(8 120 459 floats = 32.5 MB for GeoFormer) laid out in reverse registration order -- roughly the order the backward
produces them -- and cut into buckets (default 8 MB: xGMI is point-to-point, 7 links x ~153 GB/s, so a ring moves
32.5 MB in ~0.4 ms and small buckets would only add latency).  Every ``p.grad`` starts a step as ``None``, so autograd
hands a parameter's gradient over without an accumulation launch; a post-accumulate hook counts a bucket's parameters,
and the moment the last one is in, the bucket's gradients are packed into their slice of the buffer by one
concatenation, ``p.grad`` is re-pointed at the slice and the bucket's asynchronous all-reduce starts, while the rest of
the backward is still running.  Buckets always go out in index order and ``finish()`` sends the ones the backward
never completed (parameters without a gradient, e.g. decoder and heads before ``prepare_epochs``, contribute zeros),
so every rank issues the identical sequence of collectives even when the ranks' graphs differ (an empty-foreground
batch on one rank).  Parameters that received a gradient on NO rank get ``grad = None`` back, so the optimizer skips
them exactly like the single-GPU step does (no weight decay / moment updates on modules that are not trained yet).
With one process nothing is packed or exchanged: the gradients stay where autograd put them.

``SyncBatchNorm1d``: batch statistics over all ranks with ONE packed collective per layer and direction (forward: an
all-gather of every rank's (count, mean, M2) merged with Chan's parallel formula, the arithmetic of
``batch_norm_gather_stats_with_counts``; backward: an all-reduce of (sum(dy), sum(dy * xhat))).  A layer's statistics
are needed before the next convolution can run, and the layers of a U-Net level feed each other (BN -> ReLU -> conv ->
BN ...), so the collectives of different layers cannot be packed into one: SURVEY 8e's "one packed all-reduce per
level" would need statistics that are one layer stale.  What is packed is everything one layer exchanges.  A rank with
an EMPTY input still takes part (count 0), and ``convert_sync_batchnorm`` gives the model a ``rank_agreement`` hook so
that all ranks leave the forward together when one of them has no foreground -- the ranks' collective sequences stay
identical.  With SyncBatchNorm layers in the module the gradient buckets are NOT started from the backward hooks
(they would interleave with the layers' backward collectives in a rank-dependent order on the same communicator) but
all from ``finish()``.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist
import torch.nn as nn


# Run the collectives of SyncBatchNorm1d even in a process group of ONE rank (tests: the RCCL calls of the data-parallel
# step executed on the one GPU a test box has; tests/test_gpu_rccl.py).  BucketedGradReducer has its own switch.
FORCE_EXCHANGE = False


def _exchanging():
    return dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_EXCHANGE)


def init_distributed(backend=None, force=False):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  One process needs no
    group; force=True creates the one-rank group anyway (the RCCL path on a single GPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if dist.is_initialized() or (world == 1 and not force):
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if "MASTER_PORT" not in os.environ:
            import socket

            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            s.close()
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")  # "nccl" IS RCCL on ROCm
    dist.init_process_group(backend)
    return world


class BucketedGradReducer:
    """Gradient averaging over one flat buffer; buckets start from gradient hooks while the backward runs."""

    def __init__(self, module, bucket_bytes=8 << 20, only_trainable=True, overlap=None, always_exchange=False):
        """always_exchange: hooks, packing and the all-reduces also run in a process group of ONE rank (the step a
        multi-GPU job executes, on the one GPU a test has; without a process group there is nothing to call).
        overlap: start a bucket's all-reduce from the gradient hooks while the backward runs.  Default: on, unless the
        module holds SyncBatchNorm1d layers -- their backward collectives share the communicator, and a rank whose
        first bucket completes late would order the two kinds differently from its peers."""
        if overlap is None:
            overlap = not any(isinstance(m, SyncBatchNorm1d) for m in module.modules())
        self.overlap = bool(overlap)
        self.always_exchange = bool(always_exchange)
        params = [p for p in module.parameters() if (p.requires_grad or not only_trainable)]
        self.params = params[::-1]  # backward order: the last layers' gradients arrive first
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.used = [0.0] * len(self.params)  # host side; uploaded once in finish()
        per_bucket = max(int(bucket_bytes) // 4, 1)
        self.views, self.bucket_of, self.ranges = [], [], []
        off = start = 0
        for p in self.params:
            if off - start >= per_bucket:  # close the bucket before this parameter
                self.ranges.append((start, off))
                start = off
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            self.bucket_of.append(len(self.ranges))
            off += p.numel()
        self.ranges.append((start, off))
        self.nparams_in = [0] * len(self.ranges)
        self.members = [[] for _ in self.ranges]  # parameter indices of every bucket
        for i, b in enumerate(self.bucket_of):
            self.nparams_in[b] += 1
            self.members[b].append(i)
        self._hooks = []
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self.launched_in_backward = 0  # buckets whose all-reduce started before finish() (for tests / logs)
        self.prepare()

    def numel(self):
        return self.flat.numel()

    def world(self):
        return dist.get_world_size() if dist.is_initialized() else 1

    def exchanging(self):
        """Is there somebody to exchange with (or a one-rank group with always_exchange)?"""
        return dist.is_initialized() and (dist.get_world_size() > 1 or self.always_exchange)

    @torch.no_grad()
    def prepare(self):
        """Call before every backward (instead of ``zero_grad``): every ``p.grad`` is dropped, so autograd hands each
        parameter's gradient over without an accumulation launch; a bucket's gradients are packed into the flat buffer in
        one launch when the bucket is complete (``_pack``).  (Pointing the gradients at zeroed views of the buffer, as
        this class first did, costs a fill of the whole buffer plus one in-place add per parameter and step: ~400 launches.)"""
        self.used = [0.0] * len(self.params)
        for p in self.params:
            p.grad = None
        # the hooks exist only while there is somebody to exchange with (one process: ~400 Python calls per backward
        # on the autograd thread for nothing)
        if self.exchanging() and not self._hooks:
            self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        elif not self.exchanging() and self._hooks:
            self.remove_hooks()
        self._ready = [0] * len(self.ranges)
        self._next = 0
        self._works = []
        self._fired = [False] * len(self.params)
        self.launched_in_backward = 0

    def _pack(self, b):
        """The gradients of bucket b into their slice of the flat buffer (one concatenation; zeros where a parameter got
        no gradient), and the parameters' ``grad`` re-pointed at the slice so that the optimizer reads the averaged values.
        One process: nothing to exchange -- the gradients stay where autograd put them."""
        if not self.exchanging():
            return
        s, e = self.ranges[b]
        idxs = self.members[b]
        grads = [self.params[i].grad for i in idxs]
        if all(g is not None for g in grads):
            torch.cat([g.reshape(-1) for g in grads], out=self.flat[s:e])
        else:
            self.flat[s:e].zero_()
            have = [(self.views[i], g) for i, g in zip(idxs, grads) if g is not None]
            if have:
                torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for i, g in zip(idxs, grads):
            if g is not None:
                self.params[i].grad = self.views[i]

    def _launch(self, b):
        self._pack(b)
        s, e = self.ranges[b]
        if self.exchanging():
            self._works.append(dist.all_reduce(self.flat[s:e], async_op=True))

    @torch.no_grad()
    def _on_grad(self, p):
        i = self._index[id(p)]
        if self._fired[i]:
            return
        self._fired[i] = True
        self.used[i] = 1.0
        b = self.bucket_of[i]
        self._ready[b] += 1
        # buckets leave in index order only: identical collective sequence on every rank
        while self.overlap and self._next < len(self.ranges) and self._ready[self._next] == self.nparams_in[self._next]:
            self._launch(self._next)
            self._next += 1
            self.launched_in_backward += 1

    @torch.no_grad()
    def finish(self):
        """After backward: send the buckets the hooks did not complete, wait, average, and give parameters that no
        rank produced a gradient for ``grad = None`` back."""
        world = self.world()
        while self._next < len(self.ranges):
            self._launch(self._next)
            self._next += 1
        if not self.exchanging():
            return
        flags = torch.tensor(self.used, dtype=torch.float32, device=self.flat.device)
        self._works.append(dist.all_reduce(flags, op=dist.ReduceOp.MAX, async_op=True))
        for w in self._works:
            w.wait()
        self.flat.div_(world)
        used = flags.tolist()
        for p, v, u in zip(self.params, self.views, used):
            p.grad = v if u > 0 else None

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def all_ranks_agree(flag: bool, device) -> bool:
    """True iff `flag` holds on every rank (one MIN all-reduce of a word + its read-back)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float32,
                     device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


class _SyncBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        C = x.shape[1]
        red = [d for d in range(x.dim()) if d != 1]
        n_local = x.numel() // C
        if n_local > 0:
            var_l, mean_l = torch.var_mean(x, red, unbiased=False)  # two-pass / Welford: no E[x^2] - mean^2 cancellation
            m2_l = var_l * float(n_local)
        else:  # an empty rank still takes part, with count 0
            mean_l, m2_l = x.new_zeros(C), x.new_zeros(C)
        local = torch.cat([x.new_full((1,), float(n_local)), mean_l, m2_l])
        if _exchanging():
            flat = x.new_empty(dist.get_world_size() * (2 * C + 1))  # (flat: gloo takes the concatenated form only)
            dist.all_gather_into_tensor(flat, local)
            allst = flat.view(dist.get_world_size(), 2 * C + 1)
        else:
            allst = local.view(1, -1)
        # Chan et al.: n = sum n_r, mean = sum n_r mean_r / n, M2 = sum M2_r + sum n_r (mean_r - mean)^2
        n_r, mean_r, m2_r = allst[:, :1], allst[:, 1:1 + C], allst[:, 1 + C:]
        n = n_r.sum().clamp_min(1.0)
        mean = (n_r * mean_r).sum(0) / n
        var = ((m2_r + n_r * (mean_r - mean) ** 2).sum(0) / n).clamp_min(0.0)
        invstd = torch.rsqrt(var + eps)
        shape = [1, C] + [1] * (x.dim() - 2)
        xhat = (x - mean.view(shape)) * invstd.view(shape)
        ctx.save_for_backward(xhat, weight, invstd, n)
        ctx.red, ctx.shape = red, shape
        return xhat * weight.view(shape) + bias.view(shape), mean, var, n

    @staticmethod
    def backward(ctx, gy, _gm, _gv, _gn):
        xhat, weight, invstd, n = ctx.saved_tensors
        red, shape = ctx.red, ctx.shape
        C = xhat.shape[1]
        gw_local = (gy * xhat).sum(red)
        gb_local = gy.sum(red)
        packed = torch.cat([gb_local, gw_local])
        if _exchanging():
            dist.all_reduce(packed)  # statistics of the GLOBAL batch; weight/bias grads stay local (the reducer sums them)
        sum_dy, sum_dy_xhat = packed[:C], packed[C:]
        gx = (gy - sum_dy.view(shape) / n - xhat * (sum_dy_xhat.view(shape) / n)) * (weight * invstd).view(shape)
        return gx, gw_local, gb_local, None


class SyncBatchNorm1d(nn.BatchNorm1d):
    """BatchNorm1d whose training statistics span all ranks (one packed all-reduce per direction); eval mode and the
    state dict are those of nn.BatchNorm1d."""

    sync_across_ranks = True  # (the single-rank fused BatchNorm + ReLU pair of spconv.SparseSequential must not take it)

    def forward(self, x):
        if not self.training:
            return super().forward(x)
        # (an empty input goes through as well: the other ranks are waiting in this layer's collective)
        y, mean, var, n = _SyncBNFn.apply(x, self.weight, self.bias, self.eps)
        with torch.no_grad():
            if self.track_running_stats:
                m = self.momentum if self.momentum is not None else 0.1
                unbiased = var * (n / (n - 1).clamp_min(1.0))
                self.running_mean.mul_(1 - m).add_(mean, alpha=m)
                self.running_var.mul_(1 - m).add_(unbiased, alpha=m)
                self.num_batches_tracked += 1
        return y


def convert_sync_batchnorm(module):
    """Replace every nn.BatchNorm1d (incl. the build's lean subclass) by SyncBatchNorm1d, sharing parameters and
    buffers (what the reference's ``nn.SyncBatchNorm.convert_sync_batchnorm`` call, train.py:182, intended).  A model
    with a ``rank_agreement`` attribute (GeoFormer) gets ``all_ranks_agree``: its forward then leaves on ALL ranks when
    one rank's batch has no foreground, instead of skipping the heads' layers on that rank alone."""
    if hasattr(module, "rank_agreement"):
        module.rank_agreement = all_ranks_agree
    for name, child in list(module.named_children()):
        if isinstance(child, nn.BatchNorm1d) and not isinstance(child, SyncBatchNorm1d):
            sb = SyncBatchNorm1d(child.num_features, eps=child.eps, momentum=child.momentum, affine=child.affine,
                                 track_running_stats=child.track_running_stats)
            sb.weight, sb.bias = child.weight, child.bias
            sb.running_mean, sb.running_var = child.running_mean, child.running_var
            sb.num_batches_tracked = child.num_batches_tracked
            sb.train(child.training)
            setattr(module, name, sb)
        else:
            convert_sync_batchnorm(child)
    return module


def max_over_ranks(seconds: float, device=None) -> float:
    """Wall time of the slowest rank (bench.py's timing rule)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
