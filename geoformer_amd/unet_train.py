"""Host side of gf_unet_train_fwd / gf_unet_train_bwd (include/geoformer_hip.h): the sparse U-Net in TRAINING mode
(batch-statistics BatchNorm, saved activations, backward) run from native code.

The module tree (``input_conv`` -> nested ``UBlock``s -> ``output_layer``: model/geoformer/geoformer.py:39-53,398-401;
``ResidualBlock`` / ``UBlock``: geoformer_modules.py:10-35,52-129) is compiled ONCE into a list of ops over numbered
feature buffers (``Program``).  A training forward builds the step's rulebooks, carves one workspace and runs the
program in ranges: the two voxel transformers of the deepest levels stay framework modules, so a step is three ranges,
each ONE autograd function whose backward is one native call.  Every op issues exactly the launches the per-module
route makes (``spconv`` / ``pointops.bn_relu_train``), so results match that route; what disappears is the host side of
~600 framework / ctypes calls forward and ~135 Python autograd functions + ~250 framework nodes backward per step.
Everything that is computed happens in libgeoformer_hip.so; this file collects pointers.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib, pointops, sparse
from ._lib import check, stream_ptr

_FP = ctypes.c_void_p


class TrainOp(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("kind", "level", "table", "src", "dst", "aux", "Cin", "Cout", "no_dgrad", "pad_")] + \
        [(n, _FP) for n in ("w", "gamma", "beta", "running_mean", "running_var")] + \
        [("eps", ctypes.c_float), ("momentum", ctypes.c_float)] + \
        [(n, ctypes.c_longlong) for n in ("wp_off", "pgrad_off", "stats_off")]


class TrainLevel(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int), ("ld", ctypes.c_int), ("nbr", _FP), ("gmask", _FP), ("steps", _FP),
                ("M_coarse", ctypes.c_int), ("ld_down", ctypes.c_int), ("child", _FP), ("gmask_down", _FP),
                ("ld_up", ctypes.c_int), ("pad_", ctypes.c_int), ("up", _FP), ("gmask_up", _FP), ("flat", _FP)]


BN_RELU, CONV, CAT = 0, 1, 2
T_1X1, T_SUBM, T_DOWN, T_UP = 0, 1, 2, 3


def _r(n, m):
    return (int(n) + m - 1) // m * m


class Segment:
    def __init__(self, begin, in_buf):
        self.begin, self.end, self.in_buf, self.out_buf = begin, begin, in_buf, -1
        self.params, self.grads = [], []  # parameters in op order; (pgrad offset, numel, shape) of each
        self.bns = []
        self.transformer = None  # (UBlock, level) applied to the range's output by the framework


def tree_signature(model):
    """Identity of everything a Program bakes in: the ids of the U-Net's modules and the addresses of their parameters
    and buffers (~900 integers, ~0.4 ms per training forward).  A module replaced (convert_sync_batchnorm, a swapped
    block) or a tensor re-allocated (load_state_dict(assign=True), ``p.data = ...``, an EMA swap) changes it."""
    sig = []
    for root in (model.input_conv, model.unet, model.output_layer):
        for mod in root.modules():
            sig.append(id(mod))
            for t in mod._parameters.values():
                if t is not None:
                    sig.append(t.data_ptr())
            for t in mod._buffers.values():
                if t is not None:
                    sig.append(t.data_ptr())
    return tuple(sig)


class Program:
    """The module tree as ops.  Parameters and BatchNorm buffers are referenced by ADDRESS (in-place optimizer updates
    keep them valid); the program holds a reference to every such tensor (``keep``: the addresses stay allocated for as
    long as the program lives) and the tree's signature, against which every forward re-validates it (``_program``)."""

    def __init__(self, model, signature=None):
        lib = _lib.load()
        self.ops, self.bufs, self.segments = [], [], []
        self.wp_floats = self.pgrad_floats = self.stats_floats = 0
        self.keep = []
        self.signature = tree_signature(model) if signature is None else signature
        ic = model.input_conv[0]
        x0 = self._buf(0, ic.in_channels)
        self.seg = Segment(0, x0)
        x = self._conv(ic, T_SUBM, 0, x0, 0, no_dgrad=True)
        x = self._ublock(model.unet, x, 0)
        out = self._bn(model.output_layer[0], x, 0)
        self._close(out)
        self.nlevels = max(level for level, _ in self.bufs) + 1
        self.array = (TrainOp * len(self.ops))(*self.ops)
        self.ref = ctypes.addressof(self.array)
        self.buf_level = np.array([b[0] for b in self.bufs], dtype=np.int64)
        self.buf_C = np.array([b[1] for b in self.bufs], dtype=np.int64)
        self.lib = lib

    # -- compilation --------------------------------------------------------------------------
    def _buf(self, level, C):
        self.bufs.append((level, C))
        return len(self.bufs) - 1

    def _close(self, out_buf, transformer=None):
        self.seg.end, self.seg.out_buf, self.seg.transformer = len(self.ops), out_buf, transformer
        self.segments.append(self.seg)

    def _param(self, p, off, n):
        assert p.dtype == torch.float32 and p.is_contiguous() and p.requires_grad  # (on the GPU: `supported`)
        self.seg.params.append(p)
        self.seg.grads.append((off, n, tuple(p.shape)))

    def _bn(self, bn, src, level):
        C = self.bufs[src][1]
        assert isinstance(bn, nn.BatchNorm1d) and bn.num_features == C and bn.affine and bn.track_running_stats \
            and bn.momentum is not None and C % 4 == 0 and C <= 256
        dst = self._buf(level, C)
        op = TrainOp(kind=BN_RELU, level=level, src=src, dst=dst, aux=-1, Cin=C, Cout=C)
        op.gamma, op.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
        op.running_mean, op.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        self.keep += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        op.eps, op.momentum = float(bn.eps), float(bn.momentum)
        op.stats_off, op.pgrad_off = self.stats_floats, self.pgrad_floats
        self.stats_floats += 2 * C
        self._param(bn.weight, self.pgrad_floats, C)
        self._param(bn.bias, self.pgrad_floats + C, C)
        self.pgrad_floats += 2 * C
        self.seg.bns.append(bn)
        self.ops.append(op)
        return dst

    def _conv(self, conv, table, level, src, dst_level, aux=-1, no_dgrad=False):
        Cin, Cout = conv.in_channels, conv.out_channels
        K = {T_1X1: 1, T_SUBM: 27, T_DOWN: 8, T_UP: 8}[table]
        assert conv.bias is None and self.bufs[src][1] == Cin and conv.weight.numel() == K * Cin * Cout
        assert Cout % 4 == 0 and (aux < 0 or self.bufs[aux] == (dst_level, Cout))
        dst = self._buf(dst_level, Cout)
        op = TrainOp(kind=CONV, level=level, table=table, src=src, dst=dst, aux=aux, Cin=Cin, Cout=Cout,
                     no_dgrad=int(no_dgrad))
        op.w = conv.weight.data_ptr()
        self.keep.append(conv.weight)
        op.wp_off, op.pgrad_off = self.wp_floats, self.pgrad_floats
        self.wp_floats += _r(_lib.load().gf_conv_packed_floats(K, Cin, Cout), 64)
        self._param(conv.weight, self.pgrad_floats, K * Cin * Cout)
        self.pgrad_floats += _r(K * Cin * Cout, 4)
        self.ops.append(op)
        return dst

    def _block(self, blk, x, level):
        """ResidualBlock (geoformer_modules.py:10-35): conv1(relu(bn1(conv0(relu(bn0(x)))))) + i_branch(x)."""
        bn0, r0, conv0, bn1, r1, conv1 = list(blk.conv_branch._modules.values())
        assert type(r0) is nn.ReLU and type(r1) is nn.ReLU
        ib = blk.i_branch[0]
        idn = x if isinstance(ib, nn.Identity) else self._conv(ib, T_1X1, level, x, level)
        a0 = self._bn(bn0, x, level)
        c0 = self._conv(conv0, T_SUBM, level, a0, level)
        a1 = self._bn(bn1, c0, level)
        return self._conv(conv1, T_SUBM, level, a1, level, aux=idn)

    def _ublock(self, u, x, level):
        """UBlock.forward (geoformer_modules.py:99-129)."""
        for blk in u.blocks._modules.values():
            x = self._block(blk, x, level)
        if len(u.nPlanes) > 1:
            o = x
            assert type(u.conv[1]) is nn.ReLU and type(u.deconv[1]) is nn.ReLU
            ad = self._bn(u.conv[0], o, level)
            xd = self._conv(u.conv[2], T_DOWN, level, ad, level + 1)
            inner = self._ublock(u.u, xd, level + 1)
            au = self._bn(u.deconv[0], inner, level + 1)
            dec = self._conv(u.deconv[2], T_UP, level, au, level)
            Co, Cd = self.bufs[o][1], self.bufs[dec][1]
            cat = self._buf(level, Co + Cd)
            self.ops.append(TrainOp(kind=CAT, level=level, src=o, dst=cat, aux=dec, Cin=Co, Cout=Cd))
            x = cat
            for blk in u.blocks_tail._modules.values():
                x = self._block(blk, x, level)
        if u.before_transformer_linear is not None:
            # the dense per-scene transformer stays a framework module: the range ends here, the next one starts from
            # the transformer's output
            self._close(x, transformer=(u, level))
            t = self._buf(level, self.bufs[x][1])
            self.seg = Segment(len(self.ops), t)
            x = t
        return x


def _bn_list(model):
    from .unet_exec import _bn_modules

    return _bn_modules(model)


def supported(model, x, signature=None):
    """Training on the GPU with every U-Net parameter trainable and every BatchNorm in training mode.  The structural
    part of the answer is cached per tree signature: a tree that changed (SyncBatchNorm conversion after a first
    training forward, re-allocated tensors) is examined again."""
    if os.environ.get("GF_UNET_TRAIN_EXEC", "1") == "0" or os.environ.get("GF_FUSED_BN", "1") == "0":
        return False
    f = x.features
    if not (torch.is_grad_enabled() and f.is_cuda and f.dtype == torch.float32 and f.shape[0] >= 2 and not f.requires_grad):
        return False
    sig = tree_signature(model) if signature is None else signature
    hit = model.__dict__.get("_gf_unet_train_static")
    if hit is None or hit[0] != sig:
        try:
            from .model.backbone import ResidualBlock, UBlock
            from .unet_exec import _levels

            lv = _levels(model.unet)
            ok = isinstance(model.unet, UBlock) and all(
                isinstance(b, ResidualBlock) for u in lv
                for seq in [u.blocks] + ([u.blocks_tail] if len(u.nPlanes) > 1 else []) for b in seq._modules.values())
            ok = ok and all(u.nPlanes[0] % 4 == 0 and u.nPlanes[0] * 2 <= 256 for u in lv) and model.input_conv[0].bias is None
            ok = ok and type(model.output_layer[1]) is nn.ReLU
            bns = _bn_list(model) if ok else []
            ok = ok and all(isinstance(b, nn.BatchNorm1d) and type(b).__name__ != "SyncBatchNorm1d" and b.affine
                            and b.track_running_stats and b.momentum is not None for b in bns)
        except Exception:  # an unfamiliar module tree: the module route handles it
            ok, bns, lv = False, [], []
        params = [p for m in (model.input_conv, model.unet, model.output_layer) for p in m.parameters()]
        hit = model.__dict__["_gf_unet_train_static"] = (sig, ok, bns, params, len(lv))
    _, ok, bns, params, nl = hit
    if not ok or not all(b.training for b in bns) or not all(p.requires_grad for p in params):
        return False
    if not all((int(s) >> k) >= 2 for s in x.spatial_shape for k in range(nl - 1)):
        return False
    return sig  # (truthy; unet_forward takes it, so the tree is walked once per forward)


def _program(model, signature=None):
    sig = tree_signature(model) if signature is None else signature
    hit = model.__dict__.get("_gf_unet_train_prog")
    if hit is None or hit.signature != sig:
        hit = model.__dict__["_gf_unet_train_prog"] = Program(model, sig)
    return hit


class Run:
    """One training forward / backward: rulebooks, workspaces and pointer tables (alive until the backward is done)."""

    def __init__(self, prog, levels, rows, keep, device):
        self.prog, self.levels, self.keep, self.device = prog, levels, keep, device
        self.levels_ref = ctypes.addressof(levels)
        nb = len(prog.bufs)
        sizes = (np.asarray(rows, dtype=np.int64)[prog.buf_level] * prog.buf_C + 63) // 64 * 64
        self.sizes = sizes
        offs = np.concatenate([[0], np.cumsum(sizes)])
        self.offs = offs
        act_floats = int(offs[-1])
        lib = prog.lib
        self.scratch_floats = int(lib.gf_unet_train_scratch_floats(prog.ref, len(prog.ops), self.levels_ref))
        total = act_floats + prog.wp_floats + _r(prog.stats_floats, 64) + self.scratch_floats
        self.ws = torch.empty(total, dtype=torch.float32, device=device)
        base = self.ws.data_ptr()
        self.act = (ctypes.c_void_p * nb)(*[base + 4 * int(o) for o in offs[:-1]])
        self.wp_ptr = base + 4 * act_floats
        self.stats_ptr = self.wp_ptr + 4 * prog.wp_floats
        self.scratch_ptr = self.stats_ptr + 4 * _r(prog.stats_floats, 64)
        self.rows = rows
        self.gws = None
        self.grad = (ctypes.c_void_p * nb)()
        self.ghas = (ctypes.c_ubyte * nb)()
        self.pgrad = None

    def view(self, buf):
        level, C = self.prog.bufs[buf]
        o = int(self.offs[buf])
        return self.ws[o:o + self.rows[level] * C].view(self.rows[level], C)

    def ensure_grad(self):
        if self.gws is None:
            self.gws = torch.empty(int(self.offs[-1]), dtype=torch.float32, device=self.device)
            self.pgrad = torch.empty(self.prog.pgrad_floats, dtype=torch.float32, device=self.device)
            base = self.gws.data_ptr()
            for i, o in enumerate(self.offs[:-1]):
                self.grad[i] = base + 4 * int(o)

    def grad_view(self, buf):
        level, C = self.prog.bufs[buf]
        o = (int(self.grad[buf]) - self.gws.data_ptr()) // 4  # (a residual operand's gradient may have been re-pointed)
        return self.gws[o:o + self.rows[level] * C].view(self.rows[level], C)


class _SegFn(torch.autograd.Function):
    """One range of the program: forward = gf_unet_train_fwd, backward = gf_unet_train_bwd."""

    @staticmethod
    def forward(ctx, run, k, h, *params):
        seg = run.prog.segments[k]
        h = h.contiguous()
        assert h.dtype == torch.float32 and h.shape == (run.rows[run.prog.bufs[seg.in_buf][0]], run.prog.bufs[seg.in_buf][1])
        run.keep.append(h.detach())  # (detached: the tensor's history leads back to the previous range's node -> this Run)
        run.act[seg.in_buf] = h.data_ptr()
        check(run.prog.lib.gf_unet_train_fwd(run.prog.ref, seg.begin, seg.end, run.levels_ref, run.act, run.wp_ptr,
                                             run.stats_ptr, run.scratch_ptr, stream_ptr()), "gf_unet_train_fwd")
        for bn in seg.bns:  # num_batches_tracked: counted on the host (model/layers.py: _HostCounter)
            bn._nbt_pending = getattr(bn, "_nbt_pending", 0) + 1
        ctx.run, ctx.k = run, k
        return run.view(seg.out_buf)

    @staticmethod
    def backward(ctx, g):
        run, k = ctx.run, ctx.k
        seg = run.prog.segments[k]
        run.ensure_grad()
        g = g.contiguous()
        run.keep.append(g.detach())
        run.grad[seg.out_buf] = g.data_ptr()
        run.ghas[seg.out_buf] = 2  # the caller's tensor: never written into
        check(run.prog.lib.gf_unet_train_bwd(run.prog.ref, seg.begin, seg.end, run.levels_ref, run.act, run.grad, run.ghas,
                                             run.stats_ptr, run.pgrad.data_ptr(), run.scratch_ptr, stream_ptr()),
              "gf_unet_train_bwd")
        gin = run.grad_view(seg.in_buf) if k > 0 else None
        pg = [run.pgrad[o:o + n].view(shape) for o, n, shape in seg.grads]
        if k == 0:
            ctx.run = None  # the last range of the backward: workspaces go back to the allocator
        return (None, None, gin, *pg)


def unet_forward(model, x, batch_size, signature=None):
    """Output features [M,16] (with autograd history) of input_conv -> unet -> output_layer for the SparseConvTensor x
    in training mode; None when a level is too small for the native route (the module route then runs)."""
    prog = _program(model, signature)
    coords = x._coords()
    nl = prog.nlevels
    chain = sparse.down_rules_chain(coords, x.batch_size, x.spatial_shape, nl - 1)
    cur = coords
    for l, r in enumerate(chain):  # (also where the module route finds them)
        r.prebuilt_for = cur.data_ptr()
        x.indice_dict[f"spconv{l + 1}"] = r
        cur = r.out_coords
    if len(chain) != nl - 1:
        return None
    rows = [int(coords.shape[0])] + [int(r.M_out) for r in chain]
    if min(rows) < 2:
        return None
    level_coords = [coords] + [r.out_coords for r in chain]
    index = [x._level_index()] + [r.index_out for r in chain]
    keep = [chain, level_coords, index]
    levels = (TrainLevel * nl)()
    for l in range(nl):
        s = sparse.subm_rules(level_coords[l], index[l])
        keep.append(s)
        L = levels[l]
        L.M, L.ld, L.nbr, L.gmask = rows[l], s.ld, s.nbr.data_ptr(), s.gmask.data_ptr()
        L.steps = None if s.steps is None else s.steps.data_ptr()
        L.flat = None if s.flat is None else s.flat.data_ptr()
        if l < nl - 1:
            r = chain[l]
            L.M_coarse, L.ld_down, L.child, L.gmask_down = rows[l + 1], r.ld, r.child.data_ptr(), r.gmask_down.data_ptr()
            L.ld_up, L.up, L.gmask_up = r.ld_up, r.up.data_ptr(), r.gmask_up.data_ptr()
    run = Run(prog, levels, rows, keep, x.features.device)
    h = x.features
    for k, seg in enumerate(prog.segments):
        h = _SegFn.apply(run, k, h, *seg.params)
        if seg.transformer is not None:
            u, level = seg.transformer
            c = level_coords[level]
            if pointops.backbone_transformer_train_supported(h, c, u.transformer):
                # forward and backward of the stack as a handful of native launches (csrc/backbone_attn.hip) instead of
                # ~250 framework launches per level: 3.3 ms of host time per level of the batch-4 step
                h = pointops.backbone_transformer_train(h, c, batch_size, u.before_transformer_linear, u.transformer,
                                                        u.after_transformer_linear)
                continue
            feats = u.before_transformer_linear(h)
            feats = u.transformer(xyz=c[:, 1:].float(), features=feats, batch_ids=c[:, 0], batch_size=batch_size)
            h = u.after_transformer_linear(feats)
    return h
