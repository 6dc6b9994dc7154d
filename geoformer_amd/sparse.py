"""Host side of the sparse-convolution path: rulebook objects and kernel launches.

Thin Python over the C ABI (include/geoformer_hip.h).  PyTorch supplies device buffers and
the current stream; all arithmetic happens in libgeoformer_hip.so.  The objects here play
the role of spconv 1.0's ``indice_dict`` entries: one
``SubmRules`` per ``indice_key="submL"`` and one ``DownRules`` per ``indice_key="spconvL"``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def _round16(n: int) -> int:
    return (int(n) + 15) // 16 * 16


STEPS_MIN_ROWS = 6000 * 16  # gf_conv_fwd takes the counted-loop kernel from 6000 groups up (csrc/spconv_conv.hip)
FLAT_MIN_ROWS = 1500 * 16  # gf_conv_fwd_flat takes the LDS-weight kernel from 1500 groups up (csrc/spconv_lw.hip)


@dataclass
class LevelIndex:
    """Occupancy-bitmap rank index of one voxel set (see csrc/spconv_rules.hip)."""

    bitmap: torch.Tensor  # uint32 words (stored as int32)
    prefix: torch.Tensor  # int32
    perm: Optional[torch.Tensor]  # rank -> row (None when rows are in rank order)
    batch: int
    shape: tuple  # (X, Y, Z)


@dataclass
class SubmRules:
    nbr: torch.Tensor  # int32 [27, ld]
    gmask: torch.Tensor  # int32 [ld/16]
    ld: int
    M: int
    K: int = 27
    steps: Optional[torch.Tensor] = None  # int32: step table [ld/16, 7, 16, 4] + chunk boundaries (gf_rules_subm3)
    flat: Optional[torch.Tensor] = None  # int32: flat step table (gf_rules_flat_steps) for the LDS-weight conv kernel

    def pairs(self):
        """Canonical spconv pair lists: for each offset k the (in,out) pairs in ascending out."""
        out = []
        for k in range(self.K):
            col = self.nbr[k, : self.M]
            o = torch.nonzero(col >= 0).view(-1)
            out.append(torch.stack([col[o].long(), o]))
        return out


@dataclass
class DownRules:
    out_coords: torch.Tensor  # int32 [M_out,4]
    M_in: int
    M_out: int
    child: torch.Tensor  # int32 [8, ld]
    ld: int
    gmask_down: torch.Tensor
    parent: torch.Tensor  # int32 [M_in]
    koff: torch.Tensor  # int32 [M_in]
    up: torch.Tensor  # int32 [8, ld_up]
    ld_up: int
    gmask_up: torch.Tensor
    index_out: LevelIndex
    out_shape: tuple


def _scratch(words: int, device):
    lib = _lib.load()
    nbytes = lib.gf_index_scratch_bytes(words)
    return torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=device)


def build_index(coords: torch.Tensor, batch: int, shape) -> LevelIndex:
    """coords int32 [M,4] (b,x,y,z) on the GPU, unique rows."""
    lib = _lib.load()
    assert coords.is_cuda and coords.dtype == torch.int32 and coords.is_contiguous()
    X, Y, Z = (int(s) for s in shape)
    M = coords.shape[0]
    words = lib.gf_index_words(batch, X, Y, Z)
    dev = coords.device
    bitmap = torch.empty(words, dtype=torch.int32, device=dev)
    prefix = torch.empty(words, dtype=torch.int32, device=dev)
    perm = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
    scratch = _scratch(words, dev)
    check(
        lib.gf_index_build(ptr(coords), M, None, batch, X, Y, Z, ptr(bitmap), ptr(prefix), ptr(perm), ptr(scratch),
                           stream_ptr()),
        "gf_index_build",
    )
    return LevelIndex(bitmap, prefix, perm, batch, (X, Y, Z))


def subm_rules(coords: torch.Tensor, index: LevelIndex) -> SubmRules:
    lib = _lib.load()
    M = coords.shape[0]
    ld = max(_round16(M), 16)
    dev = coords.device
    nbr = torch.empty((27, ld), dtype=torch.int32, device=dev)
    gmask = torch.empty(ld // 16, dtype=torch.int32, device=dev)
    # the step table only pays where the counted-loop kernel runs (level-1 sized voxel sets)
    steps = torch.empty(lib.gf_rules_steps_words(ld), dtype=torch.int32, device=dev) if ld >= STEPS_MIN_ROWS else None
    X, Y, Z = index.shape
    check(
        lib.gf_rules_subm3(ptr(coords), M, None, X, Y, Z, ptr(index.bitmap), ptr(index.prefix), ptr(index.perm),
                           ptr(nbr), ld, ptr(gmask), ptr(steps), stream_ptr()),
        "gf_rules_subm3",
    )
    flat = flat_steps(nbr, gmask, 27, M, ld) if ld >= FLAT_MIN_ROWS else None
    return SubmRules(nbr, gmask, ld, M, 27, steps, flat)


def flat_steps(nbr: torch.Tensor, gmask: torch.Tensor, K: int, M: int, ld: int, nbins: int = 0) -> torch.Tensor:
    """Flat step table of a [K, ld] relation (include/geoformer_hip.h: gf_rules_flat_steps)."""
    lib = _lib.load()
    flat = torch.empty(lib.gf_rules_flat_words(K, ld), dtype=torch.int32, device=nbr.device)
    check(lib.gf_rules_flat_steps(ptr(nbr), ptr(gmask), K, M, ld, nbins, ptr(flat), stream_ptr()), "gf_rules_flat_steps")
    return flat


def down_rules(coords: torch.Tensor, batch: int, shape) -> DownRules:
    """k=2, s=2 rulebook; one host sync to learn the number of output voxels."""
    lib = _lib.load()
    X, Y, Z = (int(s) for s in shape)
    OX, OY, OZ = (X - 2) // 2 + 1, (Y - 2) // 2 + 1, (Z - 2) // 2 + 1
    M = coords.shape[0]
    dev = coords.device
    words = lib.gf_index_words(batch, OX, OY, OZ)
    bitmap = torch.empty(words, dtype=torch.int32, device=dev)
    prefix = torch.empty(words, dtype=torch.int32, device=dev)
    scratch = _scratch(words, dev)
    ld = max(_round16(M), 16)  # capacity: M_out <= M_in
    ld_up = ld
    out_coords = torch.empty((ld, 4), dtype=torch.int32, device=dev)
    d_M_out = torch.zeros(1, dtype=torch.int32, device=dev)
    child = torch.empty((8, ld), dtype=torch.int32, device=dev)
    parent = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
    koff = torch.empty(max(M, 1), dtype=torch.int32, device=dev)
    up = torch.empty((8, ld_up), dtype=torch.int32, device=dev)
    gmask_down = torch.empty(ld // 16, dtype=torch.int32, device=dev)
    gmask_up = torch.empty(ld_up // 16, dtype=torch.int32, device=dev)
    check(
        lib.gf_rules_down2(ptr(coords), M, None, batch, X, Y, Z, ptr(bitmap), ptr(prefix), ptr(scratch),
                           ptr(out_coords), ptr(d_M_out), ptr(child), ld, ptr(parent), ptr(koff), ptr(up), ld_up,
                           ptr(gmask_down), ptr(gmask_up), stream_ptr()),
        "gf_rules_down2",
    )
    M_out = int(d_M_out.item())
    index_out = LevelIndex(bitmap, prefix, None, batch, (OX, OY, OZ))
    return DownRules(out_coords[:M_out], M, M_out, child, ld, gmask_down, parent[:M], koff[:M], up, ld_up, gmask_up,
                     index_out, (OX, OY, OZ))


import weakref

_PACK_CACHE = {}  # id(weight tensor) -> (weakref, (ptr, version, ...), packed)


def pack_weights(weight: torch.Tensor) -> torch.Tensor:
    """[K,Cin,Cout] (or spconv's [k,k,k,Cin,Cout]) fp32 -> MFMA B-operand stream order.

    Cached per (storage, version) so eval-mode weights are packed once; a training step
    that updates the parameter in place bumps ``_version`` and triggers a re-pack."""
    lib = _lib.load()
    Cin, Cout = int(weight.shape[-2]), int(weight.shape[-1])
    K = weight.numel() // (Cin * Cout)
    key = (weight.data_ptr(), weight._version, K, Cin, Cout)
    hit = _PACK_CACHE.get(id(weight))
    if hit is not None and hit[0]() is weight and hit[1] == key:
        return hit[2]
    w = weight.detach()
    assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
    wp = torch.empty(lib.gf_conv_packed_floats(K, Cin, Cout), dtype=torch.float32, device=w.device)
    check(lib.gf_conv_pack_weights(ptr(w), K, Cin, Cout, ptr(wp), stream_ptr()), "gf_conv_pack_weights")
    wid = id(weight)
    _PACK_CACHE[wid] = (weakref.ref(weight, lambda _r, wid=wid: _PACK_CACHE.pop(wid, None)), key, wp)
    return wp


def conv_fwd(feats: torch.Tensor, weight: torch.Tensor, nbr: Optional[torch.Tensor], gmask: Optional[torch.Tensor],
             K: int, M_out: int, ld: int, in_scale=None, in_shift=None, residual=None, out=None,
             events=None, steps=None, out_scale=None, out_shift=None, packed=None, flat=None) -> torch.Tensor:
    """out[o] = sum_k act(feats[nbr[k,o]]) @ weight[k] (+ residual), optionally followed by the epilogue activation
    max(out*out_scale + out_shift, 0).  weight is [K,Cin,Cout] fp32; packed = (wp, Cin, Cout): weights that are
    packed already (weight is ignored)."""
    lib = _lib.load()
    assert feats.is_cuda and feats.dtype == torch.float32 and feats.is_contiguous()
    if packed is not None:
        Cin, Cout = int(packed[1]), int(packed[2])
    else:
        assert weight.dtype == torch.float32 and weight.is_contiguous()
        Cin, Cout = int(weight.shape[-2]), int(weight.shape[-1])
    assert feats.shape[1] == Cin, (feats.shape, weight.shape)
    if out is None:
        out = torch.empty((M_out, Cout), dtype=torch.float32, device=feats.device)
    if residual is not None:
        assert residual.is_contiguous() and residual.shape == (M_out, Cout)
    wp = packed[0] if packed is not None else pack_weights(weight)
    if events is not None:  # (start, stop) torch.cuda.Event pair recorded around the launch in native code
        from ctypes import c_void_p

        for e in events:
            if not e.cuda_event:
                e.record()  # materialise the hipEvent_t handle
        check(
            lib.gf_dev_conv_fwd_timed(ptr(feats), ptr(wp), ptr(nbr), ptr(gmask), ptr(steps), K, feats.shape[0], M_out, ld, Cin, Cout, ptr(in_scale),
                                  ptr(in_shift), ptr(residual), ptr(out_scale), ptr(out_shift), ptr(out), events[0].cuda_event,
                                  events[1].cuda_event, stream_ptr()),
            "gf_dev_conv_fwd_timed",
        )
        return out
    if flat is not None:
        check(
            lib.gf_conv_fwd_flat(ptr(feats), ptr(wp), ptr(nbr), ptr(gmask), ptr(steps), ptr(flat), K, feats.shape[0], M_out, ld,
                                 Cin, Cout, ptr(in_scale), ptr(in_shift), ptr(residual), ptr(out_scale), ptr(out_shift),
                                 ptr(out), None, stream_ptr()),
            "gf_conv_fwd_flat",
        )
        return out
    check(
        lib.gf_conv_fwd(ptr(feats), ptr(wp), ptr(nbr), ptr(gmask), ptr(steps), K, feats.shape[0], M_out, ld, Cin, Cout, ptr(in_scale),
                        ptr(in_shift), ptr(residual), ptr(out_scale), ptr(out_shift), ptr(out), stream_ptr()),
        "gf_conv_fwd",
    )
    return out


def dev_conv_knobs(split=-1, wide=-1, pair=-1, ldsw=0, block=0, g16=-1, g16_ldsw=-1, g16_gpw=0, g16_pipe=-1, flat=-1,
                   flat_items=0, lw=-1, lw_items=0):
    """Dev hook (include/geoformer_hip_dev.h): force gf_conv_fwd's launch shape; no arguments = size-based choice."""
    check(_lib.load().gf_dev_conv_knobs(split, wide, pair, ldsw, block), "gf_dev_conv_knobs")
    check(_lib.load().gf_dev_conv_knobs_g16(g16, g16_ldsw, g16_gpw, g16_pipe), "gf_dev_conv_knobs_g16")
    check(_lib.load().gf_dev_conv_knob_flat(flat, flat_items), "gf_dev_conv_knob_flat")
    check(_lib.load().gf_dev_conv_knob_lw(lw, lw_items), "gf_dev_conv_knob_lw")


def dev_conv_g16p_wpb(wpb=0):
    """Dev hook: waves per workgroup of the pipelined level-1 conv kernel (4, 8, 12, 16; 0 = default)."""
    check(_lib.load().gf_dev_conv_g16p_wpb(wpb), "gf_dev_conv_g16p_wpb")


def dev_conv_chunks(n=0):
    """Dev hook: number of equal-cost chunks the next submanifold rulebooks are built with (0 = default)."""
    check(_lib.load().gf_dev_conv_chunks(n), "gf_dev_conv_chunks")


def resblock_fwd(x: torch.Tensor, wp0, wp1, wpi, nbr, gmask, K: int, M: int, ld: int, Cin: int, Cout: int, s0, t0, s1,
                 t1, events=None, steps=None) -> torch.Tensor:
    """Eval-mode pre-activation residual block in one native call (include/geoformer_hip.h: gf_resblock_fwd).
    wp0/wp1/wpi are packed weights (pack_weights), s*/t* folded BatchNorm vectors.  events: optional two
    (start, stop) torch.cuda.Event pairs recorded natively around the two 3x3x3 launches (bench.py's probe);
    the block then goes out as separate launches of the same kernels."""
    lib = _lib.load()
    buf = torch.empty((3 if wpi is not None else 2, M, Cout), dtype=torch.float32, device=x.device)
    st = stream_ptr()
    if events is not None:
        for pair in events:
            for e in pair:
                if not e.cuda_event:
                    e.record()  # materialise the hipEvent_t handle
        idn = x
        if wpi is not None:
            check(lib.gf_conv_fwd(x.data_ptr(), wpi.data_ptr(), None, None, None, 1, M, M, 0, Cin, Cout, None, None, None,
                                  None, None, buf[2].data_ptr(), st), "gf_conv_fwd")
            idn = buf[2]
        sp = ptr(steps)
        check(lib.gf_dev_conv_fwd_timed(x.data_ptr(), wp0.data_ptr(), nbr.data_ptr(), gmask.data_ptr(), sp, K, M, M, ld, Cin,
                                    Cout, s0.data_ptr(), t0.data_ptr(), None, s1.data_ptr(), t1.data_ptr(), buf[1].data_ptr(),
                                    events[0][0].cuda_event, events[0][1].cuda_event, st), "gf_dev_conv_fwd_timed")
        check(lib.gf_dev_conv_fwd_timed(buf[1].data_ptr(), wp1.data_ptr(), nbr.data_ptr(), gmask.data_ptr(), sp, K, M, M, ld,
                                    Cout, Cout, None, None, idn.data_ptr(), None, None, buf[0].data_ptr(),
                                    events[1][0].cuda_event, events[1][1].cuda_event, st), "gf_dev_conv_fwd_timed")
        return buf[0]
    check(lib.gf_resblock_fwd(x.data_ptr(), wp0.data_ptr(), wp1.data_ptr(), None if wpi is None else wpi.data_ptr(),
                              nbr.data_ptr(), gmask.data_ptr(), ptr(steps), K, M, ld, Cin, Cout, s0.data_ptr(), t0.data_ptr(),
                              s1.data_ptr(), t1.data_ptr(), buf[1].data_ptr(),
                              None if wpi is None else buf[2].data_ptr(), buf[0].data_ptr(), st), "gf_resblock_fwd")
    return buf[0]


def conv_dgrad(grad_out: torch.Tensor, weight: torch.Tensor, bwd, M_in: int) -> torch.Tensor:
    """Input gradient = the forward kernel over the transposed relation.
    bwd = ("subm", (nbr, gmask, 27, M, ld)): same table, weights W[26-k]^T (the submanifold
    relation is symmetric: nbr[k][o] = i  <=>  nbr[26-k][i] = o);
    bwd = ("table", (tbl, gmask, K, M_in, ld)): explicit transposed table, weights W[k]^T."""
    kind, spec = bwd
    tbl, gmask, K, M, ld = spec[:5]
    steps = spec[5] if len(spec) > 5 else None
    Cin, Cout = int(weight.shape[-2]), int(weight.shape[-1])
    w = weight.detach()
    assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
    lib = _lib.load()
    wp = torch.empty(lib.gf_conv_packed_floats(K, Cout, Cin), dtype=torch.float32, device=w.device)
    check(lib.gf_conv_pack_weights_t(ptr(w), K, Cin, Cout, 1 if kind == "subm" else 0, ptr(wp), stream_ptr()),
          "gf_conv_pack_weights_t")
    return conv_fwd(grad_out, None, tbl, gmask, K, M, ld, steps=steps, packed=(wp, Cout, Cin))


def conv_wgrad(feats: torch.Tensor, grad_out: torch.Tensor, nbr, K: int, M_out: int, ld: int, gmask=None) -> torch.Tensor:
    """dW[k] = sum_o feats[nbr[k][o]]^T grad_out[o]; with the table's group masks the (group, offset) pairs that have
    no neighbour at all are skipped."""
    lib = _lib.load()
    Cin, Cout = feats.shape[1], grad_out.shape[1]
    dW = torch.empty((K, Cin, Cout), dtype=torch.float32, device=feats.device)
    if gmask is not None and nbr is not None and K <= 32:
        check(lib.gf_conv_wgrad_masked(ptr(feats), ptr(grad_out), ptr(nbr), ptr(gmask), K, M_out, ld, Cin, Cout, ptr(dW),
                                       stream_ptr()), "gf_conv_wgrad_masked")
    else:
        check(lib.gf_conv_wgrad(ptr(feats), ptr(grad_out), ptr(nbr), K, M_out, ld, Cin, Cout, ptr(dW), stream_ptr()),
              "gf_conv_wgrad")
    return dW


_PLAN_CACHE = {}


def _down_chain_plan(M0: int, batch: int, shape, nlevels: int):
    import ctypes

    key = (M0, batch, tuple(shape), nlevels)
    hit = _PLAN_CACHE.get(key)
    if hit is None:
        lib = _lib.load()
        offs = (ctypes.c_longlong * (nlevels * 10))()
        caps = (ctypes.c_int * (nlevels + 1))()
        shapes = (ctypes.c_int * (3 * (nlevels + 1)))()
        total, nl = ctypes.c_longlong(0), ctypes.c_int(0)
        check(lib.gf_rules_down2_chain_plan(M0, batch, shape[0], shape[1], shape[2], nlevels,
                                            ctypes.cast(offs, ctypes.c_void_p), ctypes.cast(caps, ctypes.c_void_p),
                                            ctypes.cast(shapes, ctypes.c_void_p), ctypes.addressof(total),
                                            ctypes.addressof(nl)), "gf_rules_down2_chain_plan")
        n = nl.value
        hit = (n, [list(offs[l * 10:(l + 1) * 10]) for l in range(n)], list(caps[: n + 1]),
               [tuple(shapes[3 * l:3 * l + 3]) for l in range(n + 1)], int(total.value))
        if len(_PLAN_CACHE) > 64:
            _PLAN_CACHE.clear()
        _PLAN_CACHE[key] = hit
    return hit


def down_rules_chain(coords: torch.Tensor, batch: int, shape, nlevels: int):
    """Rulebooks of `nlevels` successive k=2/s=2 down-samplings: one native call, one workspace, one host sync
    (include/geoformer_hip.h: gf_rules_down2_chain).  Level l+1 is built from level l's output coordinates with the
    voxel count kept on the device; tables are sized by a host-known bound of the level's voxel count and keep that
    capacity as leading dimension; the counts come back in a single D2H copy.  Returns a list of DownRules."""
    lib = _lib.load()
    dev = coords.device
    M0 = coords.shape[0]
    shape = tuple(int(s) for s in shape)
    nl, offs, caps, shapes, total = _down_chain_plan(M0, batch, shape, nlevels)
    if nl == 0:
        return []
    ws = torch.empty(total, dtype=torch.int32, device=dev)
    counts = torch.empty(nl + 1, dtype=torch.int32, device=dev)
    check(lib.gf_rules_down2_chain(coords.data_ptr(), M0, batch, shape[0], shape[1], shape[2], nl, ws.data_ptr(),
                                   counts.data_ptr(), stream_ptr()), "gf_rules_down2_chain")
    # every view that does not depend on the voxel counts is made while the chain is still running on the device;
    # after the read-back only three slices per level are left (the stretch behind the sync is launch-bound)
    pre = []
    for l in range(nl):
        o, cap_in, cap_out, oshape = offs[l], caps[l], caps[l + 1], shapes[l + 1]
        words = lib.gf_index_words(batch, *oshape)
        pre.append((ws[o[3]:o[3] + 4 * cap_out].view(cap_out, 4), ws[o[4]:o[4] + 8 * cap_out].view(8, cap_out),
                    ws[o[8]:o[8] + cap_out // 16], ws[o[5]:o[5] + cap_in], ws[o[6]:o[6] + cap_in],
                    ws[o[7]:o[7] + 8 * cap_in].view(8, cap_in), ws[o[9]:o[9] + cap_in // 16],
                    LevelIndex(ws[o[0]:o[0] + words], ws[o[1]:o[1] + words], None, batch, oshape)))
    n = [M0] + (counts[1:].tolist() if M0 > 0 else [0] * nl)  # the only host sync
    rules = []
    cin = coords
    for l in range(nl):
        oc, child, gd, parent, koff, up, gu, index = pre[l]
        r = DownRules(oc[: n[l + 1]], n[l], n[l + 1], child, caps[l + 1], gd, parent[: n[l]], koff[: n[l]], up, caps[l],
                      gu, index, shapes[l + 1])
        r.in_coords, r.in_shape = cin[: n[l]] if l else coords, list(shapes[l])
        rules.append(r)
        cin = oc
    return rules
