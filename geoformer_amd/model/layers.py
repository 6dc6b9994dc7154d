"""Pure-PyTorch building blocks of GeoFormer with the reference's parameter names.

Written from the reference's behaviour (cited per class) so that its checkpoints load by
name; everything here is stock PyTorch-ROCm plumbing around the HIP operators.
"""
from __future__ import annotations

import copy
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import pointops

_FUSED_BN = os.environ.get("GF_FUSED_BN", "1") != "0"  # dev knob: the framework's kernels for training-mode BatchNorm


def _bn_train(mod, x, dims):
    """Training-mode batch norm written with plain reductions (autograd-differentiable).  MIOpen's training
    BN costs ~1.1 ms of HOST time per call on this stack (87 calls = 97 ms of a 180 ms training forward)."""
    if _FUSED_BN and isinstance(mod, _HostCounter) and pointops.bn_train_cl_supported(mod, x):
        # the whole layer as three launches per direction (csrc/bn_train.hip) instead of var_mean + five element-wise
        # passes forward and a dozen kernels backward
        y = pointops.bn_train_cl(mod, x)
        mod._nbt_pending = getattr(mod, "_nbt_pending", 0) + 1  # counted on the host (see BatchNorm1d below)
        return y
    shape = [1, -1] + [1] * (x.dim() - 2)
    var, mean = torch.var_mean(x, dims, unbiased=False)  # one pass (Welford) instead of two reductions
    if mod.track_running_stats:
        with torch.no_grad():
            n = x.numel() // x.shape[1]
            mom = mod.momentum if mod.momentum is not None else 1.0 / float(mod.num_batches_tracked + 1)
            mod.running_mean.mul_(1 - mom).add_(mean.detach(), alpha=mom)
            mod.running_var.mul_(1 - mom).add_(var.detach() * (n / max(n - 1, 1)), alpha=mom)
            mod.num_batches_tracked += 1
    y = (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + mod.eps)
    if mod.affine:
        y = y * mod.weight.view(shape) + mod.bias.view(shape)
    return y


class _HostCounter:
    """`num_batches_tracked` counted on the host between state-dict saves (a device launch per layer and step for a
    counter that only a momentum of None reads)."""

    def _flush_counter(self):
        n = getattr(self, "_nbt_pending", 0)
        if n and self.num_batches_tracked is not None:
            self.num_batches_tracked += n
        self._nbt_pending = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush_counter()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._nbt_pending = 0
        super()._load_from_state_dict(*args, **kwargs)


class BatchNorm1d(_HostCounter, nn.BatchNorm1d):
    """nn.BatchNorm1d (same parameters, buffers and eval behaviour) with a lean training forward."""

    def forward(self, x):
        # [M,C] inputs (the sparse backbone) keep the library kernel, which is fine for that shape (measured:
        # 59.6 vs 73.5 ms for the backbone-only training step); the [B,C,L] ones go the lean way
        if self.training and x.dim() == 3 and x.numel() > 0:
            return _bn_train(self, x, [0, 2])
        if self.training and x.dim() == 2 and _FUSED_BN and pointops.bn_relu_train_supported(self, x):
            # rows [N, C] outside the sparse backbone (the semantic head over all points): the same kernels without the
            # ReLU (the next module applies it)
            return pointops.bn_relu_train(self, x.contiguous(), relu=False)
        if (self.training and x.is_cuda and x.dim() == 2 and x.shape[0] > 1 and self.track_running_stats
                and self.momentum is not None):
            # the same library kernels without nn.BatchNorm1d's `num_batches_tracked += 1` on the device (a launch
            # per layer and step for a counter that only a momentum of None reads): counted on the host, written
            # back when the state is saved
            self._nbt_pending = getattr(self, "_nbt_pending", 0) + 1
            return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, True, self.momentum,
                                self.eps)
        return super().forward(x)


class BatchNorm2d(_HostCounter, nn.BatchNorm2d):
    def forward(self, x):
        if self.training and x.numel() > 0:
            return _bn_train(self, x, [0, 2, 3])
        return super().forward(x)


class _SplitKLinearFn(torch.autograd.Function):
    """y = x W^T + b with the weight gradient computed as a batched product over row chunks.  For the decoder's
    pair tensors ([nq*nc*B, 64] = ~1M rows against a 64x64 weight) the library's dW = gy^T x GEMM runs on two
    workgroups (no split over the 1M-long reduction): 1.5 ms per layer instead of the ~0.1 ms the traffic needs."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        g2 = gy.reshape(-1, gy.shape[-1])
        if ctx.needs_input_grad[0]:
            gx = (g2 @ weight).reshape(x.shape)
        if ctx.needs_input_grad[1]:
            x2 = x.reshape(-1, x.shape[-1])
            rows = x2.shape[0]
            chunks = 256 if rows >= 256 else 1
            per = rows // chunks
            main = chunks * per  # (a row count that is not a multiple of the chunk count: the rest as one small product)
            gw = torch.bmm(g2[:main].view(chunks, per, -1).transpose(1, 2), x2[:main].view(chunks, per, -1)).sum(0)
            if main < rows:
                gw = gw + g2[main:].t() @ x2[main:]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            rows = g2.shape[0]
            if rows >= (1 << 14) and g2.is_contiguous():
                # a column sum over ~10^5 rows of a 13-wide matrix is ONE slow reduction launch (0.69 ms for the
                # semantic head's [550k, 13]): per-chunk sums first (parallel over chunks x columns), then the chunks
                chunks = 256
                per = rows // chunks
                gb = g2[:chunks * per].view(chunks, per, -1).sum(1).sum(0)
                if chunks * per < rows:
                    gb = gb + g2[chunks * per:].sum(0)
            else:
                gb = g2.sum(0)
        return gx, gw, gb


class BigLinear(nn.Linear):
    """nn.Linear (same parameters / state-dict entries) for inputs with ~10^6 rows: split-K weight gradient."""

    def forward(self, x):
        if x.is_cuda and torch.is_grad_enabled() and x.numel() // max(x.shape[-1], 1) >= (1 << 12):  # (8192 context tokens: 70 -> ~15 us)
            return _SplitKLinearFn.apply(x, self.weight, self.bias)
        return super().forward(x)


def _pointwise(w, x):
    """W x[b] for w [Co,Ci], x [B,Ci,L] as a batched product.  (``torch.matmul`` of a 2-D with a 3-D operand folds the
    batch into the rows of x^T: a transposing copy of x going in and of y coming out, 80 us each over the set
    abstraction's 16.7 M group values.)"""
    # (bmm on the broadcast weight, not matmul: matmul folds whenever the small operand requires grad -- also inside an
    # autograd function's forward, where that is only the tensor's flag)
    return torch.bmm(w.unsqueeze(0).expand(x.shape[0], -1, -1), x)


class _PointwiseSplitKFn(torch.autograd.Function):
    """y[b] = W x[b] for x [B,Cin,L] with L ~ 10^5: the weight gradient sum_b gy[b] x[b]^T reduces over L into a
    Cout x Cin tile, which the library runs on a single workgroup (0.6 ms at 16x16x240k); chunked into a batched
    product it is bandwidth-bound.  The chunks are strided views of gy / x (row stride L, batch stride L/S): the
    library takes them as they lie, no packed copies."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return _pointwise(w, x)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _pointwise(w.t(), gy)
        if ctx.needs_input_grad[1]:
            B, Co, L = gy.shape
            Ci = x.shape[1]
            S = 128
            Lc = L // S
            main = S * Lc
            gw = torch.zeros_like(w)
            if Lc > 0:
                if not gy.is_contiguous():
                    gy = gy.contiguous()
                parts = gy.new_empty((B, S, Co, Ci))
                for b in range(B):
                    a = gy[b, :, :main].view(Co, S, Lc).permute(1, 0, 2)  # [S, Co, Lc], strides (Lc, L, 1)
                    c = x[b, :, :main].view(Ci, S, Lc).permute(1, 2, 0)  # [S, Lc, Ci], strides (Lc, 1, L)
                    torch.bmm(a, c, out=parts[b])
                gw = parts.sum((0, 1))
            if main < L:
                gw = gw + torch.einsum("bol,bil->oi", gy[:, :, main:], x[:, :, main:])
        return gx, gw


class PointwiseConv1d(nn.Conv1d):
    """nn.Conv1d(kernel_size=1) evaluated as a GEMM.  Same parameters / state-dict entries; MIOpen has no
    tuned kernels for these shapes on gfx950 and falls back to naive convolutions (38 ms per weight gradient
    of a [1,16,N] conv measured in the training step), rocBLAS does not."""

    def forward(self, x):
        if self.kernel_size != (1,) or self.stride != (1,) or self.padding != (0,) or self.groups != 1:
            return super().forward(x)
        if x.is_cuda and torch.is_grad_enabled() and x.dim() == 3 and x.shape[-1] >= (1 << 15):
            y = _PointwiseSplitKFn.apply(x, self.weight[:, :, 0])
        else:
            y = _pointwise(self.weight[:, :, 0], x) if x.dim() == 3 else torch.matmul(self.weight[:, :, 0], x)
        return y if self.bias is None else y + self.bias[:, None]


class PointwiseConv2d(nn.Conv2d):
    """nn.Conv2d(kernel_size=1) as a GEMM (see PointwiseConv1d)."""

    def forward(self, x):
        if self.kernel_size != (1, 1) or self.stride != (1, 1) or self.padding != (0, 0) or self.groups != 1:
            return super().forward(x)
        b, c, h, w = x.shape
        if x.is_cuda and torch.is_grad_enabled() and h * w >= (1 << 15):
            # (the set abstraction's shared MLP over [B, C, 1024, 128] groups: the library runs the weight gradient's
            # 524 288-long reduction on a handful of workgroups, 1.1 ms per layer; chunked it is bandwidth-bound)
            y = _PointwiseSplitKFn.apply(x.reshape(b, c, h * w), self.weight[:, :, 0, 0]).reshape(b, -1, h, w)
        else:
            y = _pointwise(self.weight[:, :, 0, 0], x.reshape(b, c, h * w)).reshape(b, -1, h, w)
        return y if self.bias is None else y + self.bias[None, :, None, None]


# ------------------------------------------------------------------------------------------
# model/helper.py:43-112  GenericMLP  (children live in ``self.layers``)
# ------------------------------------------------------------------------------------------
class GenericMLP(nn.Module):
    def __init__(self, input_dim, hidden_dims, output_dim, norm_fn_name=None, activation="relu", use_conv=False,
                 dropout=None, hidden_use_bias=False, output_use_bias=True, output_use_activation=False,
                 output_use_norm=False, weight_init_name=None):
        super().__init__()
        act = {"relu": nn.ReLU, "gelu": nn.GELU}[activation]
        norm = None
        if norm_fn_name == "bn1d":
            norm = BatchNorm1d
        elif norm_fn_name == "ln":
            norm = (lambda c: nn.GroupNorm(1, c)) if use_conv else nn.LayerNorm
        elif norm_fn_name == "id":
            norm = lambda c: nn.Identity()  # noqa: E731
        elif norm_fn_name is not None:
            raise ValueError(norm_fn_name)
        if dropout is not None and not isinstance(dropout, list):
            dropout = [dropout] * len(hidden_dims)

        def lin(i, o, bias):
            return PointwiseConv1d(i, o, 1, bias=bias) if use_conv else nn.Linear(i, o, bias=bias)

        mods, prev = [], input_dim
        for i, h in enumerate(hidden_dims):
            mods.append(lin(prev, h, hidden_use_bias))
            if norm:
                mods.append(norm(h))
            mods.append(act())
            if dropout is not None:
                mods.append(nn.Dropout(p=dropout[i]))
            prev = h
        mods.append(lin(prev, output_dim, output_use_bias))
        if output_use_norm:
            mods.append(norm(output_dim))
        if output_use_activation:
            mods.append(act())
        self.layers = nn.Sequential(*mods)
        if weight_init_name == "xavier_uniform":
            for p in self.parameters():
                if p.dim() > 1:
                    nn.init.xavier_uniform_(p)

    def forward(self, x):
        return self.layers(x)


# ------------------------------------------------------------------------------------------
# model/pos_embedding.py:88-115 (fourier branch) + util/utils_pc.py:35-61 (shift_scale_points)
# ------------------------------------------------------------------------------------------
def shift_scale_points(xyz, src_range):
    """Map src_range=[lo,hi] to the unit cube: ((x - lo) * 1) / (hi - lo) + 0 in the reference; the
    multiplication by one and the addition of zero are exact, so this is bit-identical."""
    lo, hi = src_range[0][:, None, :], src_range[1][:, None, :]
    return (xyz - lo) / (hi - lo)


class PositionEmbeddingCoordsSine(nn.Module):
    """Fourier features with a frozen Gaussian projection ``gauss_B`` [3, d_pos/2]."""

    def __init__(self, temperature=10000, normalize=False, scale=None, pos_type="fourier", d_pos=None, d_in=3,
                 gauss_scale=1.0):
        super().__init__()
        if pos_type != "fourier":
            raise NotImplementedError("only the fourier embedding is used by GeoFormer (geoformer.py:119)")
        assert d_pos is not None and d_pos % 2 == 0
        self.normalize, self.pos_type, self.d_pos = normalize, pos_type, d_pos
        self.register_buffer("gauss_B", torch.empty((d_in, d_pos // 2)).normal_() * gauss_scale)

    @torch.no_grad()
    def forward(self, xyz, num_channels=None, input_range=None):
        assert xyz.ndim == 3
        d_out = (num_channels or self.d_pos) // 2
        b, n = xyz.shape[0], xyz.shape[1]
        x = xyz.clone()
        if self.normalize:
            x = shift_scale_points(x, input_range)
        x = (x * (2 * np.pi)).float()
        proj = torch.mm(x.view(-1, x.shape[-1]), self.gauss_B[:, :d_out]).view(b, n, d_out)
        return torch.cat([proj.sin(), proj.cos()], dim=2).permute(0, 2, 1)  # batch x d_pos x n


# ------------------------------------------------------------------------------------------
# model/transformer.py:153-188  backbone TransformerEncoder (levels 6 and 7 of the U-Net)
# ------------------------------------------------------------------------------------------
class Norm(nn.Module):
    """alpha * (x - mean) / (std_unbiased + eps) + bias   (transformer.py:62-76)"""

    def __init__(self, d_model, eps=1e-6):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(d_model))
        self.bias = nn.Parameter(torch.zeros(d_model))
        self.eps = eps

    def forward(self, x):
        mu = x.mean(dim=-1, keepdim=True)
        return self.alpha * (x - mu) / (x.std(dim=-1, keepdim=True) + self.eps) + self.bias


class _MHA(nn.Module):
    def __init__(self, heads, d_model, dropout=0.1):
        super().__init__()
        self.h, self.d_k, self.d_model = heads, d_model // heads, d_model
        self.q_linear = nn.Linear(d_model, d_model)
        self.v_linear = nn.Linear(d_model, d_model)
        self.k_linear = nn.Linear(d_model, d_model)
        self.dropout = nn.Dropout(dropout)
        self.out = nn.Linear(d_model, d_model)

    def forward(self, q, k, v, mask=None):
        bs = q.size(0)
        k = self.k_linear(k).view(bs, -1, self.h, self.d_k).transpose(1, 2)
        q = self.q_linear(q).view(bs, -1, self.h, self.d_k).transpose(1, 2)
        v = self.v_linear(v).view(bs, -1, self.h, self.d_k).transpose(1, 2)
        s = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(self.d_k)
        if mask is not None:
            s = s.masked_fill(mask.unsqueeze(1) == 0, -1e9)
        s = self.dropout(F.softmax(s, dim=-1))
        o = torch.matmul(s, v).transpose(1, 2).contiguous().view(bs, -1, self.d_model)
        return self.out(o)


class _FF(nn.Module):
    def __init__(self, d_model, d_ff=64, dropout=0.1):
        super().__init__()
        self.linear_1 = nn.Linear(d_model, d_ff)
        self.dropout = nn.Dropout(dropout)
        self.linear_2 = nn.Linear(d_ff, d_model)

    def forward(self, x):
        return self.linear_2(self.dropout(F.relu(self.linear_1(x))))


class _EncLayer(nn.Module):
    def __init__(self, d_model, heads, d_ff, dropout=0.1):
        super().__init__()
        self.norm_1, self.norm_2 = Norm(d_model), Norm(d_model)
        self.attn_1 = _MHA(heads, d_model)
        self.ff = _FF(d_model, d_ff=d_ff)
        self.dropout_1, self.dropout_2 = nn.Dropout(dropout), nn.Dropout(dropout)

    def forward(self, x, mask=None):
        x2 = self.norm_1(x)
        x = x + self.dropout_1(self.attn_1(x2, x2, x2, mask))
        x2 = self.norm_2(x)
        return x + self.dropout_2(self.ff(x2))


def scene_counts(batch_ids, batch_size):
    """Rows per scene, int64 [batch_size], without a device round trip: torch.bincount reads the maximum back to size
    its output even when minlength is given (3 ms of blocked host per call in the batch-4 training step)."""
    ids = batch_ids.view(-1, 1)
    return (ids == torch.arange(batch_size, device=batch_ids.device, dtype=batch_ids.dtype)).sum(0)


class BackboneTransformer(nn.Module):
    """Per scene: x = features + Linear3->d(mean_j(xyz_i - xyz_j)); N pre-norm layers; Norm."""

    def __init__(self, d_model, N, heads, d_ff):
        super().__init__()
        self.d_model, self.N = d_model, N
        self.layers = nn.ModuleList([copy.deepcopy(_EncLayer(d_model, heads, d_ff)) for _ in range(N)])
        self.norm = Norm(d_model)
        self.position_linear = nn.Linear(3, d_model)

    def forward(self, xyz, features, batch_ids, batch_size=None):
        out = torch.zeros_like(features)
        if batch_size == 1 and features.shape[0] > 0:
            # single scene: every row belongs to it -- no device round trip to find the row range
            bounds = [0, features.shape[0]]
        else:
            # rows of a scene are contiguous: ONE read-back gives every scene's range (a nonzero + min + max per scene
            # blocked the host thirteen times per call)
            nb = batch_size if batch_size is not None else int(batch_ids.max().item()) + 1
            counts = scene_counts(batch_ids, nb)
            ends = torch.cumsum(counts, 0)
            bounds = [0] + ends.tolist()
            if nb > 1 and features.is_cuda and features.shape[0] > 0:
                return self._forward_padded(xyz, features, batch_ids, bounds, ends - counts)
        nb = len(bounds) - 1
        for b in range(nb):
            s, e = bounds[b], bounds[b + 1]
            if e == s:
                continue
            rows = slice(s, e)
            pts = xyz[s:e].view(-1, 3)
            rel = (pts.unsqueeze(1) - pts.unsqueeze(0)).float().mean(dim=1)
            x = (features[s:e].view(-1, self.d_model) + self.position_linear(rel)).unsqueeze(0)
            for layer in self.layers:
                x = layer(x, mask=None)
            out[rows] = self.norm(x).squeeze(0)
        return out

    def _forward_padded(self, xyz, features, batch_ids, bounds, starts):
        """Several scenes in ONE pass: rows padded to the longest scene, padded keys masked out (the reference
        layers' own `mask` argument: scores of masked keys are filled with -1e9, their soft-max weight is exactly 0).
        The per-scene loop above issues the ~45 small launches of a layer once per scene -- and as many again in the
        backward; on a few hundred voxels per scene the host's launch rate is all that takes time."""
        nb, d = len(bounds) - 1, self.d_model
        M = features.shape[0]
        lens = [bounds[b + 1] - bounds[b] for b in range(nb)]
        L = max(lens)
        rels = []
        for b in range(nb):  # the position term with the reference's own expression (same rounding), per scene
            pts = xyz[bounds[b]:bounds[b + 1]].view(-1, 3)
            if pts.shape[0]:
                rels.append((pts.unsqueeze(1) - pts.unsqueeze(0)).float().mean(dim=1))
        x_all = features.view(-1, d) + self.position_linear(torch.cat(rels))
        bid = batch_ids.long()
        slot = torch.arange(M, device=features.device) - starts[bid] + bid * L  # row -> padded position
        x = x_all.new_zeros(nb * L, d).index_copy(0, slot, x_all).view(nb, L, d)
        mask = torch.zeros(nb * L, device=features.device, dtype=torch.long).index_fill_(0, slot, 1).view(nb, 1, L)
        for layer in self.layers:
            x = layer(x, mask=mask)
        return self.norm(x).view(nb * L, d)[slot]


# ------------------------------------------------------------------------------------------
# model/transformer_detr.py:91-166, 345-463  DETR-style decoder with relative vector attention
# ------------------------------------------------------------------------------------------
class RelPosSpec:
    """What the fused cross-attention kernel needs instead of the materialised [nq,nc,B,d] relative
    embedding: the gathered geodesic distances and the ingredients of the fourier map."""

    def __init__(self, geo_ctx, max_geo, query_locs, context_locs, lo, hi, gauss_B):
        self.geo_ctx, self.max_geo, self.query_locs, self.context_locs = geo_ctx, max_geo, query_locs, context_locs
        self.lo, self.hi, self.gauss_B = lo, hi, gauss_B


class LazyRelPos:
    """A RelPosSpec that is built on first use: the fused inference decoder asks for it only after its first token
    stage and its context-side product are queued, so those launches do not wait for the geodesic search (building the
    spec joins the stream the search runs on)."""

    def __init__(self, fn):
        self.fn, self.value = fn, None

    def get(self):
        if self.value is None:
            self.value = self.fn()
        return self.value


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead=4, dim_feedforward=256, dropout=0.1, dropout_attn=None, activation="relu",
                 normalize_before=True, use_rel=False, norm_fn_name="ln"):
        super().__init__()
        if not use_rel or not normalize_before or norm_fn_name != "ln":
            raise NotImplementedError("GeoFormer uses the pre-norm relative-position layer (geoformer.py:122-129)")
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout, inplace=True)
        self.dropout2 = nn.Dropout(dropout, inplace=True)
        self.dropout3 = nn.Dropout(dropout, inplace=True)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        # the reference builds this one in place too (transformer_detr.py:373), which current autograd rejects
        # (it overwrites the ReLU output that ReLU's backward needs); out of place draws the same mask
        self.dropout = nn.Dropout(dropout, inplace=False)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.activation = {"relu": nn.ReLU, "gelu": nn.GELU}[activation]()
        self.nhead, self.use_rel, self.normalize_before = nhead, use_rel, normalize_before
        self.attn_mlp = nn.Sequential(BigLinear(d_model, d_model), nn.ReLU(), BigLinear(d_model, d_model))
        self.v_mlp = nn.Sequential(BigLinear(d_model, d_model))
        self.out_mlp = nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU())

    def cross_attention(self, tgt2, memory, relative_pos):
        """Vector attention (transformer_detr.py:443-454): per-channel softmax over the contexts of
        MLP(q_i - k_j + r_ij) / sqrt(d), values Linear(k_j + r_ij).  tgt2 [nq,B,d], memory [nc,B,d],
        relative_pos [nq,nc,B,d] -> [nq,B,d]."""
        if isinstance(relative_pos, RelPosSpec):
            return self._cross_attention_fused(tgt2, memory, relative_pos)
        sim = self.attn_mlp(tgt2[:, None] - memory[None] + relative_pos)
        if sim.is_cuda and sim.dtype == torch.float32:
            from .. import pointops

            attn = pointops.softmax_dim1(sim, 1.0 / float(np.sqrt(sim.shape[-1])))  # streaming kernel, fwd + bwd
        else:
            attn = F.softmax(sim / np.sqrt(sim.shape[-1]), dim=1)
        v2 = self.v_mlp(memory[None] + relative_pos)
        return (attn * v2).sum(dim=1)

    def _cross_attention_fused(self, tgt2, memory, rp):
        """Inference path on the GPU: one HIP kernel per layer (csrc/decoder_attn.hip); the parts of the two
        MLPs that do not depend on the (query, context) pair are hoisted out as small GEMMs."""
        from .. import pointops

        w1, w2, wv = self.attn_mlp[0], self.attn_mlp[2], self.v_mlp[0]
        key = (w1.weight._version, w2.weight._version, wv.weight._version, w1.weight.data_ptr())
        if getattr(self, "_wpack_key", None) != key:
            self._wpack = pointops.decoder_pack_weights(w1.weight.detach().contiguous(), w2.weight.detach().contiguous(),
                                                        wv.weight.detach().contiguous())
            self._wpack_key = key
        if torch.is_grad_enabled():
            # training: the same kernel forward (keeping only the soft-max statistics) and a fused recompute-based
            # backward for the hoisted projections and the pair weights (csrc/decoder_attn.hip)
            Q1 = w1(tgt2.clone()).permute(1, 0, 2)  # (a copy: the layer's in-place dropout2 overwrites tgt2 later)
            # (split-K weight gradient: the library runs dW = gy^T memory, a [64, nc*B] x [nc*B, 64] product, on a
            # handful of workgroups -- 70 us per layer for 67 MFLOP)
            K1 = _SplitKLinearFn.apply(memory, w1.weight, None).permute(1, 0, 2)
            Kv = wv(memory).permute(1, 0, 2)
            out = pointops.decoder_cross_attn_train(rp.geo_ctx, rp.max_geo, rp.query_locs, rp.context_locs, rp.lo, rp.hi,
                                                    rp.gauss_B, Q1, K1, Kv, w1.weight, w2.weight, wv.weight)
            return out.permute(1, 0, 2)
        Q1 = w1(tgt2).permute(1, 0, 2).contiguous()  # B x nq x d
        K1 = F.linear(memory, w1.weight).permute(1, 0, 2).contiguous()  # B x nc x d
        Kv = wv(memory).permute(1, 0, 2).contiguous()
        out = pointops.decoder_cross_attn(rp.geo_ctx, rp.max_geo, rp.query_locs, rp.context_locs, rp.lo, rp.hi,
                                          rp.gauss_B, Q1, K1, Kv, self._wpack, w2.bias.detach().contiguous())
        return out.permute(1, 0, 2)  # nq x B x d

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None, relative_pos=None,
                return_attn_weights=False):
        tgt2 = self.norm1(tgt)
        q = k = tgt2 if query_pos is None else tgt2 + query_pos
        tgt2 = self.self_attn(q, k, value=tgt2, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)[0]
        tgt = tgt + self.dropout1(tgt2)
        tgt2 = self.norm2(tgt)
        tgt = self.out_mlp(self.cross_attention(tgt2, memory, relative_pos))
        tgt = tgt + self.dropout2(tgt2)  # the residual is the NORMED query (transformer_detr.py:457)
        tgt2 = self.norm3(tgt)
        tgt2 = self.linear2(self.dropout(self.activation(self.linear1(tgt2))))
        tgt = tgt + self.dropout3(tgt2)
        return tgt, None


def _decoder_fused_cache(dec):
    """Per-decoder constants of the fused inference path: pointer tables, stacked K-side weights of the
    cross-attentions, packed pair weights; rebuilt when a parameter is replaced or updated in place."""
    from .. import pointops

    params = dec.__dict__.get("_gf_fused_params")
    if params is None:
        params = dec.__dict__["_gf_fused_params"] = list(dec.parameters())
    key = (params[0].data_ptr(), sum([p._version for p in params]))
    hit = dec.__dict__.get("_gf_fused")
    if hit is not None and hit["key"] == key:
        return hit
    with torch.no_grad():
        tables = [pointops.decoder_stage_tables(l, dec.norm) for l in dec.layers]
        # K1_l = memory W1_l^T (no bias), Kv_l = memory Wv_l^T + bv_l: one batched product for all layers
        wt = torch.stack([w for l in dec.layers for w in (l.attn_mlp[0].weight.t(), l.v_mlp[0].weight.t())]).contiguous()
        bs = torch.stack([b for l in dec.layers
                          for b in (torch.zeros_like(l.v_mlp[0].bias), l.v_mlp[0].bias)]).unsqueeze(1).contiguous()
        packs = [pointops.decoder_pack_weights(l.attn_mlp[0].weight.detach().contiguous(),
                                               l.attn_mlp[2].weight.detach().contiguous(),
                                               l.v_mlp[0].weight.detach().contiguous()) for l in dec.layers]
        b2 = [l.attn_mlp[2].bias.detach().contiguous() for l in dec.layers]
    hit = {"key": key, "tables": tables, "wt": wt, "bs": bs, "packs": packs, "b2": b2}
    dec.__dict__["_gf_fused"] = hit
    return hit


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm_fn_name="ln", return_intermediate=False,
                 weight_init_name="xavier_uniform"):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(decoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = nn.LayerNorm(self.layers[0].linear2.out_features) if norm_fn_name == "ln" else None
        self.return_intermediate = return_intermediate
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def _forward_fused(self, tgt, memory, query_pos, rp):
        """Inference on the GPU: L cross-attention launches + L+1 token-stage launches + one batched product
        (csrc/decoder_attn.hip, csrc/decoder_layer.hip) instead of ~35 launches per layer."""
        from .. import pointops

        c = _decoder_fused_cache(self)
        nq, B, d = tgt.shape
        nc = memory.shape[0]
        L = len(self.layers)
        ff = self.layers[0].linear1.out_features
        mem = memory.permute(1, 0, 2).reshape(1, B * nc, d)
        kk = torch.baddbmm(c["bs"], mem.expand(2 * L, B * nc, d), c["wt"])  # [2L, B*nc, d]
        state = pointops.decoder_token_state(nq, B, tgt.device)
        inter = torch.empty((L, nq, B, d), dtype=torch.float32, device=tgt.device)
        q1 = torch.empty((B, nq, d), dtype=torch.float32, device=tgt.device)
        tgt_c, qp = tgt.contiguous(), query_pos.contiguous()
        pointops.decoder_token_stage(None, tgt_c, qp, nq, B, 4, ff, None, c["tables"][0][1], state, None, q1)
        if isinstance(rp, LazyRelPos):  # (everything above needs no geodesic distance)
            rp = rp.get()
        for l in range(L):
            attn = pointops.decoder_cross_attn(rp.geo_ctx, rp.max_geo, rp.query_locs, rp.context_locs, rp.lo, rp.hi,
                                               rp.gauss_B, q1, kk[2 * l].view(B, nc, d), kk[2 * l + 1].view(B, nc, d),
                                               c["packs"][l], c["b2"][l])
            pre = c["tables"][l + 1][1] if l + 1 < L else None
            pointops.decoder_token_stage(attn, None, qp, nq, B, 4, ff, c["tables"][l][0], pre, state, inter[l],
                                         q1 if pre is not None else None)
        return inter

    def _train_fused_dropout(self):
        """The one dropout probability of the layers (0 in eval mode), or None when they differ (framework route)."""
        ps = set()
        for l in self.layers:
            ps |= {float(l.self_attn.dropout), float(l.dropout1.p), float(l.dropout2.p), float(l.dropout3.p),
                   float(l.dropout.p)}
            if not isinstance(l.activation, nn.ReLU) or not isinstance(l.norm1, nn.LayerNorm):
                return None
        if len(ps) != 1:
            return None
        return ps.pop() if self.layers[0].self_attn.training else 0.0

    def _forward_train_fused(self, tgt, memory, query_pos, rp, p):
        """Training on the GPU: per layer the token-side stages as native forward / backward launches
        (csrc/decoder_layer_train.hip) around the fused cross-attention -- ~20 launches per layer and step instead of
        ~95.  The dropout masks are hashed from one seed per call, drawn from the framework's CPU generator."""
        from .. import pointops

        x = tgt.permute(1, 0, 2).contiguous()  # [B, nq, d] token rows
        qp = query_pos.permute(1, 0, 2).contiguous()
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        inter = []
        # the context side of every layer's cross-attention (K1_l = memory W1_l^T, Kv_l = memory Wv_l^T + bv_l) as ONE
        # product with the stacked weights: a forward GEMM and a backward pair instead of two each per layer, and the
        # gradient towards the context features is one product instead of eight accumulated ones
        d = memory.shape[-1]
        wk = torch.cat([w for layer in self.layers for w in (layer.attn_mlp[0].weight, layer.v_mlp[0].weight)], 0)
        bk = torch.cat([b for layer in self.layers for b in (torch.zeros_like(layer.v_mlp[0].bias), layer.v_mlp[0].bias)])
        kk = _SplitKLinearFn.apply(memory, wk, bk).split(d, dim=-1)  # 2 L pieces [nc, B, d]
        for l, layer in enumerate(self.layers):
            pre, post = pointops.decoder_stage_tensors(layer, self.norm)
            w1, w2, wv = layer.attn_mlp[0], layer.attn_mlp[2], layer.v_mlp[0]
            t2n, q1 = pointops.decoder_pre_train(x, qp, l, p, seed, pre)
            K1 = kk[2 * l].permute(1, 0, 2)
            Kv = kk[2 * l + 1].permute(1, 0, 2)
            ca = pointops.decoder_cross_attn_train(rp.geo_ctx, rp.max_geo, rp.query_locs, rp.context_locs, rp.lo, rp.hi,
                                                   rp.gauss_B, q1, K1, Kv, w1.weight, w2.weight, wv.weight)
            x, it = pointops.decoder_post_train(ca, t2n, l, p, seed, post)
            inter.append(it)
        return torch.stack(inter).permute(0, 2, 1, 3)  # [L, nq, B, d]

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None, relative_pos=None, transpose_swap=False,
                return_attn_weights=False):
        if isinstance(relative_pos, LazyRelPos) and (torch.is_grad_enabled() or self.layers[0].self_attn.training):
            relative_pos = relative_pos.get()
        shapes_ok = (isinstance(relative_pos, (RelPosSpec, LazyRelPos)) and tgt_mask is None and tgt_key_padding_mask is None
                     and query_pos is not None and self.norm is not None and self.return_intermediate
                     and tgt.shape[-1] == 64 and self.layers[0].nhead == 4
                     and self.layers[0].linear1.out_features % 16 == 0 and self.layers[0].linear1.out_features <= 256)
        if shapes_ok and not self.layers[0].self_attn.training and not torch.is_grad_enabled():
            return self._forward_fused(tgt, memory, query_pos, relative_pos)
        if isinstance(relative_pos, LazyRelPos):
            relative_pos = relative_pos.get()
        if (shapes_ok and torch.is_grad_enabled() and tgt.is_cuda and tgt.dtype == torch.float32
                and os.environ.get("GF_FUSED_DECODER_TRAIN", "1") != "0"):
            p = self._train_fused_dropout()
            if p is not None:
                return self._forward_train_fused(tgt, memory, query_pos, relative_pos, p)
        output, inter = tgt, []
        for layer in self.layers:
            output, _ = layer(output, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                              tgt_key_padding_mask=tgt_key_padding_mask,
                              memory_key_padding_mask=memory_key_padding_mask, pos=pos, query_pos=query_pos,
                              relative_pos=relative_pos)
            if self.return_intermediate:
                inter.append(self.norm(output))
        if self.norm is not None:
            output = self.norm(output)
            if self.return_intermediate:
                inter[-1] = output
        return torch.stack(inter) if self.return_intermediate else output
