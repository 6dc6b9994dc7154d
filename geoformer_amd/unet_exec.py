"""Host side of gf_unet_fwd (include/geoformer_hip.h): the eval-mode sparse U-Net as one native call.

Builds the parameter structs from the build's modules (``input_conv``, ``unet`` = nested ``UBlock``s, ``output_layer``:
model/geoformer/geoformer.py:42-53; geoformer_modules.py:10-35,52-129) once per parameter version and hands the call
a workspace, a pinned word per level for the voxel counts and two streams.  Everything that is computed happens in
libgeoformer_hip.so; this file only collects pointers.
"""
from __future__ import annotations

import ctypes
import threading

import torch
import torch.nn as nn

from . import _lib, sparse
from ._lib import check, stream_ptr

MAX_LEVELS = 8
_FP = ctypes.c_void_p


class ResBlockParams(ctypes.Structure):
    _fields_ = [(n, _FP) for n in ("wp0", "wp1", "wpi", "s0", "t0", "s1", "t1")]


class LevelParams(ctypes.Structure):
    _fields_ = [("C", ctypes.c_int), ("tr_layers", ctypes.c_int), ("blocks", ResBlockParams * 2),
                ("tail", ResBlockParams * 2), ("down_wp", _FP), ("down_s", _FP), ("down_t", _FP), ("up_wp", _FP),
                ("up_s", _FP), ("up_t", _FP), ("tr_params", _FP)]


class UnetParams(ctypes.Structure):
    _fields_ = [("nlevels", ctypes.c_int), ("cin", ctypes.c_int), ("input_wp", _FP), ("out_s", _FP), ("out_t", _FP),
                ("level", LevelParams * MAX_LEVELS)]


def _levels(unet):
    out = []
    u = unet
    while u is not None:
        out.append(u)
        u = getattr(u, "u", None) if len(u.nPlanes) > 1 else None
    return out


def _bn_modules(model):
    mods = [model.output_layer[0]]
    for u in _levels(model.unet):
        seqs = [u.blocks] + ([u.blocks_tail] if len(u.nPlanes) > 1 else [])
        for seq in seqs:
            for blk in seq._modules.values():
                cb = list(blk.conv_branch._modules.values())
                mods += [cb[0], cb[3]]
        if len(u.nPlanes) > 1:
            mods += [u.conv[0], u.deconv[0]]
    return mods


class UnetPlan:
    """Parameter structs of one model + the tensors they point into (kept alive here)."""

    def __init__(self, model):
        from .model.backbone import bn_affine
        from . import pointops

        keep = []

        def dp(t):
            keep.append(t)
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0
            return t.data_ptr()

        def block(blk, rb):
            bn0, _, conv0, bn1, _, conv1 = list(blk.conv_branch._modules.values())
            ib = blk.i_branch[0]
            rb.wp0, rb.wp1 = dp(sparse.pack_weights(conv0.weight)), dp(sparse.pack_weights(conv1.weight))
            rb.wpi = None if isinstance(ib, nn.Identity) else dp(
                sparse.pack_weights(ib.weight.view(1, ib.in_channels, ib.out_channels)))
            (s0, t0), (s1, t1) = bn_affine(bn0), bn_affine(bn1)
            rb.s0, rb.t0, rb.s1, rb.t1 = dp(s0), dp(t0), dp(s1), dp(t1)

        P = UnetParams()
        levels = _levels(model.unet)
        assert len(levels) <= MAX_LEVELS
        P.nlevels = len(levels)
        ic = model.input_conv[0]
        P.cin = ic.in_channels
        w16 = torch.nn.functional.pad(ic.weight.detach().reshape(27, ic.in_channels, ic.out_channels),
                                      (0, 0, 0, 16 - ic.in_channels)).contiguous()
        keep.append(w16)
        P.input_wp = dp(sparse.pack_weights(w16))
        s, t = bn_affine(model.output_layer[0])
        P.out_s, P.out_t = dp(s), dp(t)
        for l, u in enumerate(levels):
            L = P.level[l]
            L.C = u.nPlanes[0]
            blks = list(u.blocks._modules.values())
            assert len(blks) == 2, "gf_unet_fwd is laid out for block_reps = 2"
            for i, b in enumerate(blks):
                block(b, L.blocks[i])
            if len(u.nPlanes) > 1:
                for i, b in enumerate(u.blocks_tail._modules.values()):
                    block(b, L.tail[i])
                s, t = bn_affine(u.conv[0])
                L.down_wp, L.down_s, L.down_t = dp(sparse.pack_weights(u.conv[2].weight)), dp(s), dp(t)
                s, t = bn_affine(u.deconv[0])
                L.up_wp, L.up_s, L.up_t = dp(sparse.pack_weights(u.deconv[2].weight)), dp(s), dp(t)
            if u.before_transformer_linear is not None:
                table, nl = pointops.backbone_transformer_params(u.before_transformer_linear, u.transformer,
                                                                 u.after_transformer_linear)
                keep.append(table)
                L.tr_layers = nl
                L.tr_params = ctypes.cast(table, ctypes.c_void_p)
        self.params, self.keep = P, keep
        self.ref = ctypes.addressof(P)


def _version_key(model):
    ts = model.__dict__.get("_gf_unet_tensors")
    if ts is None:
        ts = [p for m in (model.input_conv, model.unet, model.output_layer) for p in list(m.parameters()) + list(m.buffers())]
        model.__dict__["_gf_unet_tensors"] = ts
    return (ts[0].data_ptr(), sum([t._version for t in ts]))


def supported(model, voxel_feats, spatial_shape):
    """Inference on the GPU with frozen BatchNorm statistics, the widths and depth gf_unet_fwd is laid out for, and a
    grid every down-sampling level exists on."""
    if torch.is_grad_enabled() or not voxel_feats.is_cuda or voxel_feats.shape[0] == 0:
        return False
    hit = model.__dict__.get("_gf_unet_static")
    if hit is None:
        ic = model.input_conv[0]
        lv = _levels(model.unet)
        ok = (ic.in_channels <= 16 and ic.out_channels == 16 and ic.bias is None and len(lv) <= MAX_LEVELS
              and all(u.nPlanes[0] % 16 == 0 for u in lv) and all(len(u.blocks) == 2 for u in lv)
              and all(u.before_transformer_linear is None or i > 0 for i, u in enumerate(lv)))
        hit = model.__dict__["_gf_unet_static"] = (ok, _bn_modules(model) if ok else [], len(lv))
    if not hit[0] or any(m.training for m in hit[1]):
        return False
    return all((int(s) >> k) >= 2 for s in spatial_shape for k in range(hit[2] - 1))


_tls = threading.local()
_SIDE_STREAMS = {}  # (device, caller stream) -> the executor's side stream


_BETWEEN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int)


def phase_next_forward(gate_events, between):
    """The NEXT unet_forward of this host thread runs as gf_unet_fwd_phased: its convolutions wait for the recorded
    torch.cuda.Event objects in gate_events, and ``between()`` -- a callable that returns a list of recorded events -- is
    called on the host once the first two levels are queued; the rest of the backbone waits for the events it returns
    (geoformer_amd/serving.py).  ``take_phase()`` tells whether a forward consumed the request."""
    _tls.phase = (list(gate_events or ()), between)


def phase_pending():
    """Whether a phase request is waiting for the next unet_forward of this host thread."""
    return getattr(_tls, "phase", None) is not None


def take_phase():
    """The pending phase request (gate events, callable) that no unet_forward has consumed, or None; clears it."""
    req = getattr(_tls, "phase", None)
    _tls.phase = None
    return req


def side_stream_for(dev, main=None):
    """The executor's side stream of the caller's current stream (created on first use)."""
    main = main or torch.cuda.current_stream(dev)
    side = _SIDE_STREAMS.get((dev, main.cuda_stream))
    if side is None:
        side = _SIDE_STREAMS[(dev, main.cuda_stream)] = torch.cuda.Stream(device=dev)
    return side


def coords_ready_next(events):
    """The NEXT unet_forward of this host thread runs as gf_unet_fwd_ahead: its coordinates wait for the recorded
    torch.cuda.Event objects in `events` (possibly none) and for nothing else the caller's stream has queued, so the
    rulebooks are built on the side stream right away -- in a loop of forwards under the previous scene's sampling / BFS
    stretch.  ``take_coords_ready()`` clears a request no forward consumed."""
    _tls.coords_ready = list(events or ())


def take_coords_ready():
    req = getattr(_tls, "coords_ready", None)
    _tls.coords_ready = None
    return req


def unet_forward(model, voxel_feats, coords, batch_size, spatial_shape):
    """[M,16] output features of input_conv -> unet -> output_layer for voxel features [M,cin] / coords int32 [M,4]."""
    lib = _lib.load()
    key = _version_key(model)
    hit = model.__dict__.get("_gf_unet_plan")
    if hit is None or hit[0] != key:
        hit = model.__dict__["_gf_unet_plan"] = (key, UnetPlan(model))
    plan = hit[1]
    M = voxel_feats.shape[0]
    X, Y, Z = (int(s) for s in spatial_shape)
    dev = voxel_feats.device
    nbytes = lib.gf_unet_ws_bytes(plan.ref, M, batch_size, X, Y, Z)
    from .pointops import scratch

    ws = scratch("unet_ws", nbytes + 256, torch.uint8, dev)  # (grow-only per stream: pointops.scratch)
    ws = ws[(-ws.data_ptr()) % 256:][:nbytes]  # the executor wants 256-byte alignment
    out = torch.empty((M, 16), dtype=torch.float32, device=dev)
    pinned = getattr(_tls, "counts", None)
    if pinned is None:
        pinned = _tls.counts = torch.zeros(MAX_LEVELS + 1, dtype=torch.int32).pin_memory()
    main = torch.cuda.current_stream(dev)
    side = side_stream_for(dev, main)  # process-wide (geoformer._SIDE_STREAMS: why)
    req = take_phase()
    ahead = take_coords_ready()
    # A workspace block that is NEW to this stream may alias memory the framework's allocator handed back while kernels of
    # the caller's stream that use it are still queued (the allocator relies on stream order): only a block this stream has
    # been using all along may be written AHEAD of the stream's queue.
    fresh = getattr(_tls, "ws_ptrs", None)
    if fresh is None:
        fresh = _tls.ws_ptrs = {}
    ws_key = (dev.index, main.cuda_stream)
    ws_is_new = fresh.get(ws_key) != ws.data_ptr()
    fresh[ws_key] = ws.data_ptr()
    if ahead is not None and (req is not None or ws_is_new):
        main.wait_stream(side)  # (the call keeps the stream's own order: the coordinates' copy on the side stream first)
        ahead = None
    if req is None and ahead is not None:
        evs = (ctypes.c_void_p * max(len(ahead), 1))(*[ctypes.c_void_p(e.cuda_event) for e in ahead])
        rc = lib.gf_unet_fwd_ahead(plan.ref, voxel_feats.data_ptr(), coords.data_ptr(), M, batch_size, X, Y, Z, ws.data_ptr(),
                                   nbytes, pinned.data_ptr(), out.data_ptr(), stream_ptr(), side.cuda_stream, evs, len(ahead))
    elif req is None:
        rc = lib.gf_unet_fwd(plan.ref, voxel_feats.data_ptr(), coords.data_ptr(), M, batch_size, X, Y, Z, ws.data_ptr(),
                             nbytes, pinned.data_ptr(), out.data_ptr(), stream_ptr(), side.cuda_stream)
    else:
        gate, between = req
        state = {"called": False, "error": None, "keep": None}

        def _between(_user, events_out, max_events):
            state["called"] = True
            try:
                evs = list(between() or ())
                if len(evs) > max_events:
                    raise RuntimeError(f"gf_unet_fwd_phased: {len(evs)} hand-over events (at most {max_events})")
                state["keep"] = evs  # alive until the call has queued its waits
                for i, e in enumerate(evs):
                    events_out[i] = e.cuda_event
                return len(evs)
            except BaseException as e:  # noqa: BLE001  (re-raised by the caller below: never through the C frame)
                state["error"] = e
                return -1

        cb = _BETWEEN(_between)
        evs = (ctypes.c_void_p * max(len(gate), 1))(*[ctypes.c_void_p(e.cuda_event) for e in gate])
        rc = lib.gf_unet_fwd_phased(plan.ref, voxel_feats.data_ptr(), coords.data_ptr(), M, batch_size, X, Y, Z,
                                    ws.data_ptr(), nbytes, pinned.data_ptr(), out.data_ptr(), stream_ptr(),
                                    side.cuda_stream, evs, len(gate), cb, None)
        if state["error"] is not None:
            side.synchronize()
            raise state["error"]
        if not state["called"]:  # an early error return: the hand-over still has to happen (the loop depends on it)
            for e in (between() or ()):
                main.wait_event(e)
    if rc != 0:
        # an error return can leave the rulebook chain queued on the side stream, still writing `ws`: the caching
        # allocator must not hand the block to the caller's stream before that work has drained
        side.synchronize()
    check(rc, "gf_unet_fwd")
    return out
