"""Host side of the point-set operators (PG_OP / pointnet2._ext / geodesic stage).

Plain functions over device tensors; every one launches a kernel of libgeoformer_hip.so on
the current stream.  The module-shaped mirrors of the reference bindings live in
``geoformer_amd.dropin``.
"""
from __future__ import annotations

import contextlib
import os
import threading

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def _f32c(t, name):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError(f"{name}: expected a contiguous float32 tensor on the GPU")
    return t


def _i32c(t, name):
    if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
        raise RuntimeError(f"{name}: expected a contiguous int32 tensor on the GPU")
    return t


# ---- PG_OP ---------------------------------------------------------------------------
def voxelize_fp(feats, rules, mode=4, out=None):
    _f32c(feats, "feats"); _i32c(rules, "rules")
    M, C = rules.shape[0], feats.shape[1]
    if out is None:
        out = torch.empty((M, C), dtype=torch.float32, device=feats.device)
    check(_lib.load().gf_voxelize_fp(ptr(feats), ptr(rules), M, rules.shape[1] - 1, C, int(mode == 4), ptr(out),
                                     stream_ptr()), "gf_voxelize_fp")
    return out


def voxelize_bp(d_out, rules, mode, d_feats):
    """d_feats [N,C] must be zero-initialised by the caller (accumulated into)."""
    _f32c(d_out, "d_out"); _i32c(rules, "rules"); _f32c(d_feats, "d_feats")
    M, C = d_out.shape
    check(_lib.load().gf_voxelize_bp(ptr(d_out), ptr(rules), M, rules.shape[1] - 1, C, int(mode == 4), ptr(d_feats),
                                     stream_ptr()), "gf_voxelize_bp")
    return d_feats


# ---- pointnet2._ext --------------------------------------------------------------------
def voxelize_idx(coords, mode=4):
    """GPU voxelisation with the semantics of PG_OP.voxelize_idx: coords int64 [N,3|4] on the device ->
    (output_coords int64 [M,ncol], input_map int32 [N], output_map int32 [M,1+maxActive]).  One host read-back
    (M, maxActive)."""
    if not (coords.is_cuda and coords.dtype == torch.int64 and coords.is_contiguous() and coords.dim() == 2):
        raise RuntimeError("coords: expected a contiguous int64 [N,3|4] tensor on the GPU")
    lib = _lib.load()
    N, ncol = coords.shape
    dev = coords.device
    scratch = torch.empty(lib.gf_voxelize_idx_scratch_bytes(N) // 8 + 1, dtype=torch.int64, device=dev)
    input_map = torch.empty(N, dtype=torch.int32, device=dev)
    head = torch.empty(3, dtype=torch.int32, device=dev)
    check(lib.gf_voxelize_idx_count(ptr(coords), N, ncol, int(mode), ptr(scratch), ptr(input_map), ptr(head),
                                    stream_ptr()), "gf_voxelize_idx_count")
    M, max_active, err = head.tolist()
    if err:
        raise _lib.GeoFormerHipError("gf_voxelize_idx: a coordinate lies outside [0, 65535] (packed 16-bit key fields)")
    max_active = max(max_active, 1)
    out_coords = torch.empty((M, ncol), dtype=torch.int64, device=dev)
    out_map = torch.empty((M, max_active + 1), dtype=torch.int32, device=dev)
    check(lib.gf_voxelize_idx_fill(ptr(coords), N, ncol, int(mode), ptr(scratch), ptr(input_map), M, max_active,
                                   ptr(out_coords), ptr(out_map), stream_ptr()), "gf_voxelize_idx_fill")
    return out_coords, input_map, out_map


def gather_points(points, idx):
    _f32c(points, "points"); _i32c(idx, "idx")
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.empty((b, c, m), dtype=torch.float32, device=points.device)
    check(_lib.load().gf_gather_points(ptr(points), ptr(idx), b, c, n, m, ptr(out), stream_ptr()), "gf_gather_points")
    return out


def gather_points_grad(grad_out, idx, n):
    _f32c(grad_out, "grad_out"); _i32c(idx, "idx")
    b, c, m = grad_out.shape
    out = torch.zeros((b, c, n), dtype=torch.float32, device=grad_out.device)
    check(_lib.load().gf_gather_points_grad(ptr(grad_out), ptr(idx), b, c, n, m, ptr(out), stream_ptr()),
          "gf_gather_points_grad")
    return out


def group_points(points, idx):
    _f32c(points, "points"); _i32c(idx, "idx")
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = torch.empty((b, c, npoints, nsample), dtype=torch.float32, device=points.device)
    check(_lib.load().gf_group_points(ptr(points), ptr(idx), b, c, n, npoints, nsample, ptr(out), stream_ptr()),
          "gf_group_points")
    return out


def group_points_grad(grad_out, idx, n):
    _f32c(grad_out, "grad_out"); _i32c(idx, "idx")
    b, c, npoints, nsample = grad_out.shape
    out = torch.zeros((b, c, n), dtype=torch.float32, device=grad_out.device)
    check(_lib.load().gf_group_points_grad(ptr(grad_out), ptr(idx), b, c, n, npoints, nsample, ptr(out),
                                           stream_ptr()), "gf_group_points_grad")
    return out


def ball_query(new_xyz, xyz, radius, nsample, grid=None):
    """pointnet2 ball_query.  grid: use the hash-grid kernel (same rows; default: for one point set of >= 4096 points)."""
    _f32c(new_xyz, "new_xyz"); _f32c(xyz, "xyz")
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=xyz.device)
    lib = _lib.load()
    if grid is None:
        grid = b == 1 and n >= 4096
    if grid and b == 1 and n >= 1:
        scratch = torch.empty(lib.gf_knn_scratch_bytes(n) // 4 + 16, dtype=torch.int32, device=xyz.device)
        check(lib.gf_ball_query_grid(ptr(xyz), n, None, ptr(new_xyz), m, float(radius), nsample, ptr(scratch), 0, None,
                                     ptr(idx), stream_ptr()), "gf_ball_query_grid")
        return idx
    check(lib.gf_ball_query(ptr(new_xyz), ptr(xyz), b, n, m, float(radius), nsample, ptr(idx), stream_ptr()),
          "gf_ball_query")
    return idx


def furthest_point_sampling(xyz, m, known=None):
    """FPS indices [b,m] (int32).  known: the first picks [b,m0] of the same sequence from an earlier call with a
    smaller m -- the draw continues from there (same result as one call with m)."""
    _f32c(xyz, "xyz")
    b, n, _ = xyz.shape
    lib = _lib.load()
    idx = torch.empty((b, m), dtype=torch.int32, device=xyz.device)
    scratch = torch.empty(lib.gf_fps_scratch_bytes(b) // 8 + 1, dtype=torch.int64, device=xyz.device)
    m0 = 0
    if known is not None:
        _i32c(known, "known")
        m0 = min(int(known.shape[1]), m)
        idx[:, :m0] = known[:, :m0]
    check(lib.gf_furthest_point_sampling_resume(ptr(xyz), b, n, m, m0, ptr(idx), ptr(scratch), stream_ptr()),
          "gf_furthest_point_sampling")
    return idx


def select_foreground(scores, cls, equal, locs, batch_idxs, feats, feat_rows=None, deferred=False):
    """Fused foreground selection (csrc/foreground.hip): points whose arg-max class is >= cls (== cls with `equal`).
    Returns (fg_idxs int64 [n], locs_ [n,3], batch_idxs_ int32 [n], feats_ [n,F], scores_ [n,C]) -- views of
    capacity-N buffers -- after ONE read-back (the count)."""
    _f32c(scores, "scores"); _f32c(locs, "locs"); _i32c(batch_idxs, "batch_idxs"); _f32c(feats, "feats")
    if feat_rows is not None:
        _i32c(feat_rows, "feat_rows")  # feats are rows of another table (voxels), read as feats[feat_rows[p]]
    N, C = scores.shape
    F = feats.shape[1]
    dev = scores.device
    lib = _lib.load()
    scratch = torch.empty(lib.gf_fg_scratch_bytes(N) // 4 + 1, dtype=torch.int32, device=dev)
    fg = torch.empty(N, dtype=torch.int64, device=dev)
    locs_o = torch.empty((N, 3), dtype=torch.float32, device=dev)
    bidx_o = torch.empty(N, dtype=torch.int32, device=dev)
    feats_o = torch.empty((N, F), dtype=torch.float32, device=dev)
    scores_o = torch.empty((N, C), dtype=torch.float32, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    host = PendingForeground.take_word() if deferred else None  # (pinned, set to -1: the scan kernel stores the count there)
    check(lib.gf_fg_select(ptr(scores), N, C, int(cls), int(bool(equal)), ptr(locs), ptr(batch_idxs), ptr(feats),
                           ptr(feat_rows), F,
                           ptr(scratch), ptr(fg), ptr(locs_o), ptr(bidx_o), ptr(feats_o), ptr(scores_o), ptr(cnt),
                           ptr(host), stream_ptr()), "gf_fg_select")
    if not deferred:
        n = int(cnt.item())
        return fg[:n], locs_o[:n], bidx_o[:n], feats_o[:n], scores_o[:n]
    return PendingForeground(cnt, (fg, locs_o, bidx_o, feats_o, scores_o), host)


class PendingForeground:
    """select_foreground(..., deferred=True): the launches are queued; ``wait()`` returns the count, ``views()`` the five
    slices.  The count reaches the host through a pinned word the scan kernel stores it to (gf_fg_select's h_count): no copy
    command, no event, and BEFORE the gathers of the outputs have run -- the host polls the word natively
    (gf_host_wait_word, interpreter lock released) and starts on what follows the count while the device finishes the
    selection.  (The staggered serving loop queues a scene's selection with its backbone and comes back for the count after
    it has queued the next scene's first launches.)"""

    _pinned = []
    _lock = threading.Lock()

    @staticmethod
    def take_word():
        with PendingForeground._lock:
            host = PendingForeground._pinned.pop() if PendingForeground._pinned else None
        if host is None:
            host = torch.zeros(1, dtype=torch.int32).pin_memory()
        host[0] = -1
        return host

    def __init__(self, cnt, bufs, host=None):
        self.polled = host is not None
        if host is None:  # (no polled word: the count as an asynchronous copy behind the selection, waited for by event)
            host = PendingForeground.take_word()
            host.copy_(cnt, non_blocking=True)
        self.host, self.bufs, self.cnt = host, bufs, cnt
        self.done = torch.cuda.Event()  # behind the whole selection (what other streams wait for before they read it)
        self.done.record()

    def wait(self):
        """Block until the count is on the host; returns it.  (``views()`` afterwards: the five slices -- a caller with
        something urgent to launch from the count alone does that in between.)"""
        import time

        t = time.perf_counter()
        if self.polled:
            n = int(_lib.load().gf_host_wait_word(self.host.data_ptr(), -1, 5_000_000))
            if n < 0:  # (five seconds without the store: the event and a plain read-back)
                self.done.synchronize()
                n = int(self.cnt.item())
        else:
            self.done.synchronize()
            n = int(self.host[0])
        _lib.host_wait_s[0] += time.perf_counter() - t
        self.n = n
        with PendingForeground._lock:
            PendingForeground._pinned.append(self.host)
        self.host = None
        return self.n

    def views(self):
        return tuple(b[:self.n] for b in self.bufs)

    def get(self):
        self.wait()
        return self.views()


_LEGACY_STATE = {}  # id(bit generator) -> (generator, address of its {uint32 key[624]; int pos}) or (generator, None)


def _legacy_state():
    """numpy's global legacy generator in place: (address of its MT19937 state -- ``uint32 key[624]; int pos``,
    numpy/random/src/mt19937/mt19937.h -- , the generator's lock), or None when the global RandomState runs on another
    bit generator or the layout does not check out against ``get_state()`` (then: the get_state / set_state route).
    ``np.random.get_state()`` + ``set_state()`` are 20-40 us each -- on the forward's critical path, once per scene."""
    import ctypes

    import numpy as np

    bg = np.random.mtrand._rand._bit_generator
    hit = _LEGACY_STATE.get(id(bg))
    if hit is None or hit[0] is not bg:
        addr = None
        if type(bg).__name__ == "MT19937":
            try:
                a = int(bg.ctypes.state_address)
                st = np.random.get_state()
                key = np.ctypeslib.as_array((ctypes.c_uint32 * 624).from_address(a))
                if st[0] == "MT19937" and int(ctypes.c_int.from_address(a + 2496).value) == int(st[2]) and \
                        bool((key == np.asarray(st[1], dtype=np.uint32)).all()):
                    addr = a
            except Exception:  # noqa: BLE001  (an interface this numpy does not have: the portable route)
                addr = None
        _LEGACY_STATE.clear()
        hit = _LEGACY_STATE[id(bg)] = (bg, addr)
    return None if hit[1] is None else (hit[1], bg.lock)


def legacy_prefetch(nwords):
    """Draw the next `nwords` outputs of numpy's global legacy generator ahead (into a per-thread native buffer, the
    generator itself untouched): a legacy_choice that starts from the same state takes its words from there."""
    import ctypes

    import numpy as np

    direct = _legacy_state()
    if direct is not None:
        addr, lock = direct
        with lock:
            check(_lib.load().gf_host_legacy_prefetch(addr, int(ctypes.c_int.from_address(addr + 2496).value), int(nwords)),
                  "gf_host_legacy_prefetch")
        return
    st = np.random.get_state()
    if st[0] != "MT19937":
        return
    key = np.ascontiguousarray(st[1], dtype=np.uint32)
    check(_lib.load().gf_host_legacy_prefetch(key.ctypes.data, int(st[2]), int(nwords)), "gf_host_legacy_prefetch")


_DRAW_PINS = {}  # (device index, stream) -> pinned int32 buffer of the fused draw (grow-only)


def draw_sample_buffers(kmax, n_cap, device, fps_m=0):
    """Device outputs and the pinned staging buffer of ``draw_sample`` for at most `kmax` drawn indices out of at most
    `n_cap` points, allocated ahead (the caller does this while it waits for the count the draw depends on).
    fps_m > 0: also the outputs of a first furthest-point-sampling launch of fps_m picks over the drawn points."""
    key = (device.index, stream_ptr())
    pin = _DRAW_PINS.get(key)
    if pin is None or pin.numel() < max(n_cap, kmax):
        pin = _DRAW_PINS[key] = torch.empty(max(n_cap + n_cap // 4, kmax, 65536), dtype=torch.int32).pin_memory()
    bufs = {"pin": pin, "idx32": torch.empty(kmax, dtype=torch.int32, device=device),
            "idx64": torch.empty(kmax, dtype=torch.int64, device=device),
            "xyz": torch.empty((1, kmax, 3), dtype=torch.float32, device=device), "kmax": kmax, "fps_m": 0}
    if fps_m > 0:
        bufs["fps_m"] = int(fps_m)
        bufs["fps_idx"] = torch.empty((1, int(fps_m)), dtype=torch.int32, device=device)
        bufs["fps_scratch"] = torch.empty(_lib.load().gf_fps_scratch_bytes(1) // 8 + 1, dtype=torch.int64, device=device)
    return bufs


def draw_sample(n, k, xyz_src, bufs):
    """The reference's per-scene draw ``np.random.choice(n, k, replace=False)`` (geoformer.py:575-577) and what follows it
    -- upload of the indices, gather of the drawn points -- as ONE native call (gf_host_draw_sample, csrc/host_draw.hip):
    returns (sampling_indices int64 [k], xyz [1,k,3]) on the device, queued on the current stream; numpy's global
    generator advances exactly as by the draw.  None when the generator cannot be driven in place (the caller then takes
    legacy_choice's route).  xyz_src: [>= n, 3] fp32 contiguous.  With buffers made for a first sampling launch
    (``fps_m``) and k >= fps_m that launch is queued in the same call: a third result, its picks int32 [1, fps_m]."""
    n, k = int(n), int(k)
    direct = _legacy_state() if 1 <= k <= n <= 0x7fffffff and k <= bufs["kmax"] else None
    if direct is None:
        return None
    _f32c(xyz_src, "xyz_src")
    addr, lock = direct
    pin = bufs["pin"]
    fps_m = bufs["fps_m"] if 0 < bufs["fps_m"] <= k else 0
    with lock:
        check(_lib.load().gf_host_draw_sample(addr, addr + 2496, n, k, pin.data_ptr(), pin.numel(), ptr(bufs["idx32"]),
                                              ptr(bufs["idx64"]), ptr(xyz_src), ptr(bufs["xyz"]), fps_m,
                                              ptr(bufs["fps_idx"]) if fps_m else None,
                                              ptr(bufs["fps_scratch"]) if fps_m else None, stream_ptr()),
              "gf_host_draw_sample")
    if fps_m:
        return bufs["idx64"][:k], bufs["xyz"][:, :k], bufs["fps_idx"]
    return bufs["idx64"][:k], bufs["xyz"][:, :k]


def legacy_choice(n, k, out=None):
    """``np.random.choice(n, k, replace=False)`` on numpy's global legacy generator -- same values, same generator
    state afterwards -- through the native restatement (csrc/host_draw.hip; about half the host time).
    out: optional int64 numpy array of at least k entries to draw into (e.g. the view of a pinned tensor, so that the
    upload that follows is an asynchronous copy); the first k entries are returned."""
    import ctypes

    import numpy as np

    n, k = int(n), int(k)
    direct = _legacy_state() if 1 <= k <= n <= 0x7fffffff else None
    if direct is not None:  # the generator's state advanced in place, under its lock
        addr, lock = direct
        if out is None or out.dtype != np.int64 or out.size < k or not out.flags.c_contiguous:
            out = np.empty(k, dtype=np.int64)
        with lock:
            check(_lib.load().gf_host_legacy_choice(addr, addr + 2496, n, k, out.ctypes.data), "gf_host_legacy_choice")
        return out[:k]
    st = np.random.get_state()
    if st[0] != "MT19937" or not (1 <= k <= n <= 0x7fffffff):
        return np.random.choice(n, k, replace=False)  # numpy's own argument errors / exotic sizes
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = ctypes.c_int32(int(st[2]))
    if out is None or out.dtype != np.int64 or out.size < k or not out.flags.c_contiguous:
        out = np.empty(k, dtype=np.int64)
    check(_lib.load().gf_host_legacy_choice(key.ctypes.data, ctypes.addressof(pos), n, k, out.ctypes.data),
          "gf_host_legacy_choice")
    np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))
    return out[:k]


# ---- geodesic stage --------------------------------------------------------------------
def knn_radius(xyz, k, radius, sqrt_out=True, check_overflow=False, return_flag=False):
    """Radius-limited kNN graph of one scene.  Returns D [n,k] fp32, I [n,k] int32, deg [n] int32
    (+ with return_flag the kernel's device-side truncation flag, int32 [1], for a caller that folds it into a
    read-back it makes anyway; check_overflow reads it here and raises)."""
    _f32c(xyz, "xyz")
    n = xyz.shape[0]
    lib = _lib.load()
    dev = xyz.device
    D = torch.empty((n, k), dtype=torch.float32, device=dev)
    I = torch.empty((n, k), dtype=torch.int32, device=dev)
    deg = torch.empty(n, dtype=torch.int32, device=dev)
    scratch = torch.empty(lib.gf_knn_scratch_bytes(n) // 4 + 16, dtype=torch.int32, device=dev)
    check(lib.gf_knn_radius(ptr(xyz), n, k, float(radius), int(sqrt_out), ptr(D), ptr(I), ptr(deg), ptr(scratch),
                            stream_ptr()), "gf_knn_radius")
    if check_overflow:
        off = (lib.gf_knn_error_flag(ptr(scratch), n) - scratch.data_ptr()) // 4
        if int(scratch[off].item()) != 0:
            raise _lib.GeoFormerHipError("gf_knn_radius: a point has more in-radius neighbours than the kernel's "
                                         "candidate list holds (rows truncated)")
    if return_flag:
        off = (lib.gf_knn_error_flag(ptr(scratch), n) - scratch.data_ptr()) // 4
        return D, I, deg, scratch[off:off + 1]
    return D, I, deg


_SCRATCH = {}


def scratch(name, numel, dtype, device):
    """Grow-only scratch tensor per (device, current stream, name): PURE scratch of one native call (workspaces the
    kernels fill and nobody reads afterwards), reused by the next call on the same stream, which stream order places
    behind this one.  The big per-scene workspaces (U-Net workspace, BFS keys and queues: ~0.8 GB for a 150k-point
    scene) then never go through the framework's caching allocator, whose blocks get split by scenes of other sizes
    until a larger scene needs a fresh hipMalloc in the middle of a forward (20-30 ms; bench.py secondary.fresh_scenes)."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream,
           name, dtype)
    t = _SCRATCH.get(key)
    if t is None or t.numel() < numel:
        _SCRATCH[key] = None  # (drop the old block before asking for the new one)
        t = _SCRATCH[key] = torch.empty(int(numel * 1.25) + 64, dtype=dtype, device=dev)
    return t[:numel]


def release_scratch(device=None):
    """Drop the grow-only scratch blocks (of one device, or of all): ~0.8 GB per stream that ran a 150k-point scene
    (U-Net workspace, BFS keys and queues, mask-head split).  Nothing else frees them -- `torch.cuda.empty_cache()` does
    not see tensors that are still referenced.  Called by `invalidate_fused_caches()` of the models and by a serving
    loop's teardown; the caller makes sure no kernel that uses them is still queued (synchronise first)."""
    if device is not None:
        dev = torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
    for key in list(_SCRATCH):
        if device is None or key[0] == idx:
            del _SCRATCH[key]


def scratch_bytes():
    """Bytes the scratch blocks hold right now (all devices and streams)."""
    return sum(t.numel() * t.element_size() for t in _SCRATCH.values() if t is not None)


def geodesic_bfs(D, I, deg, src, radius, max_step, wg_threads=1024):
    """geo [nq,n] fp32 for the sources `src` (int32 [nq]) over the kNN rows D/I (column 0 skipped).
    wg_threads: 1024 = one query per compute unit; 256 / 512 = several per unit (to run beside another kernel)."""
    _f32c(D, "D"); _i32c(I, "I"); _i32c(src, "src")
    n, K = D.shape
    nq = src.shape[0]
    dev = D.device
    geo = torch.empty((nq, n), dtype=torch.float32, device=dev)
    lib = _lib.load()
    keys = scratch("bfs_keys", nq * n, torch.int64, dev)
    qwords = int(lib.gf_geodesic_bfs_queue_words(n))
    queues = scratch("bfs_queues", nq * qwords, torch.int32, dev)
    check(lib.gf_geodesic_bfs_cfg(ptr(D), ptr(I), ptr(deg), n, K, ptr(src), nq, float(radius), int(max_step),
                                          ptr(geo), ptr(keys), ptr(queues), qwords, int(wg_threads), stream_ptr()),
          "gf_geodesic_bfs")
    return geo


# ---- fused heads -------------------------------------------------------------------------
def _mask_head_split_ws(N, device, split):
    """Scratch for the features' three bf16 pieces (gf_mask_head_episodes: fp32-accurate products on the bf16 matrix
    pipe); split=False keeps the fp32 MFMA."""
    if not split:
        return None
    return scratch("mask_head_split", _lib.load().gf_mask_head_split_bytes(N) // 8 + 1, torch.int64, device)


def mask_head(feat, coords, geo, qxyz, sqrt_max_geo, w1, b1, w2, b2, split=True):
    """Fused dynamic-conv mask head: logits [nq,N].  feat [N,16], coords [N,3], geo [nq,N] or None,
    qxyz [nq,3], sqrt_max_geo [nq] or None, w1 [nq,16,19], b1 [nq,16], w2 [nq,16], b2 [nq]."""
    for t, name in ((feat, "feat"), (coords, "coords"), (qxyz, "qxyz"), (w1, "w1"), (b1, "b1"), (w2, "w2"), (b2, "b2")):
        _f32c(t, name)
    N, C = feat.shape
    nq = qxyz.shape[0]
    out = torch.empty((nq, N), dtype=torch.float32, device=feat.device)
    check(_lib.load().gf_mask_head_episodes(ptr(feat), ptr(coords), ptr(geo), ptr(qxyz), ptr(sqrt_max_geo), ptr(w1), ptr(b1),
                                            ptr(w2), ptr(b2), 0, N, nq, 1, C, ptr(_mask_head_split_ws(N, feat.device, split)),
                                            ptr(out), stream_ptr()), "gf_mask_head")
    return out


def mask_head_packed(feat, coords, geo, qxyz, sqrt_max_geo, params, split=True):
    """mask_head with the per-query parameters read in place from the controller's output params [nq, 16*19+16+16+1]
    (column blocks w1 | w2 | b1 | b2, parse_dynamic_params of geoformer.py:264-284)."""
    return mask_head_episodes(feat, coords, geo, qxyz, sqrt_max_geo, params.unsqueeze(0), split=split)[0]


def mask_head_episodes(feat, coords, geo, qxyz, sqrt_max_geo, params, split=True):
    """E episodes over one scene in ONE launch (gf_mask_head_episodes): params [E, nq, 337] -> logits [E, nq, N];
    feat / coords / geo [nq,N] / qxyz [nq,3] / sqrt_max_geo [nq] are the scene's and shared by the episodes."""
    for t, name in ((feat, "feat"), (coords, "coords"), (qxyz, "qxyz"), (params, "params")):
        _f32c(t, name)
    N, C = feat.shape
    E, nq, ld = params.shape
    if ld != C * (C + 3) + C + C + 1 or qxyz.shape[0] != nq:
        raise RuntimeError(f"mask_head_episodes: params {tuple(params.shape)} for {qxyz.shape[0]} queries, C={C}")
    base = params.data_ptr()
    o_w2, o_b1, o_b2 = C * (C + 3), C * (C + 3) + C, C * (C + 3) + 2 * C
    out = torch.empty((E, nq, N), dtype=torch.float32, device=feat.device)
    check(_lib.load().gf_mask_head_episodes(ptr(feat), ptr(coords), ptr(geo), ptr(qxyz), ptr(sqrt_max_geo), base,
                                            base + 4 * o_b1, base + 4 * o_w2, base + 4 * o_b2, ld, N, nq, E, C,
                                            ptr(_mask_head_split_ws(N, feat.device, split)), ptr(out), stream_ptr()),
          "gf_mask_head_episodes")
    return out


class _MaskHeadFn(torch.autograd.Function):
    """Fused mask head with a fused, recompute-based backward (csrc/mask_head.hip): gradients for the mask features
    and the generated per-query parameters; coordinates and geodesic distances are data."""

    @staticmethod
    def forward(ctx, feat, params, coords, geo, qxyz, sqrt_max_geo):
        ctx.save_for_backward(feat, params, coords, geo, qxyz, sqrt_max_geo)
        return mask_head_packed(feat, coords, geo, qxyz, sqrt_max_geo, params)

    @staticmethod
    def backward(ctx, gout):
        feat, params, coords, geo, qxyz, mx = ctx.saved_tensors
        N, C = feat.shape
        nq, ld = params.shape
        lib = _lib.load()
        dparams = torch.empty_like(params)
        dfeat = torch.zeros_like(feat)
        scratch = torch.empty(lib.gf_mask_head_bwd_scratch_floats(N, nq), dtype=torch.float32, device=feat.device)
        base = params.data_ptr()
        o_w2, o_b1 = C * (C + 3), C * (C + 3) + C
        check(lib.gf_mask_head_bwd(ptr(feat), ptr(coords), ptr(geo), ptr(qxyz), ptr(mx), base, base + 4 * o_b1,
                                   base + 4 * o_w2, ptr(gout.contiguous()), ld, N, nq, C, ptr(dparams), ptr(dfeat),
                                   ptr(scratch), stream_ptr()), "gf_mask_head_bwd")
        return dfeat, dparams, None, None, None, None


class _MaskHeadEpisodesFn(torch.autograd.Function):
    """E parameter sets over one scene (the decoder layers of a training step): one forward launch and one backward
    triple for all of them; the features' gradient is summed over the episodes inside the kernel."""

    @staticmethod
    def forward(ctx, feat, params, coords, geo, qxyz, sqrt_max_geo):
        ctx.save_for_backward(feat, params, coords, geo, qxyz, sqrt_max_geo)
        return mask_head_episodes(feat, coords, geo, qxyz, sqrt_max_geo, params)

    @staticmethod
    def backward(ctx, gout):
        feat, params, coords, geo, qxyz, mx = ctx.saved_tensors
        N, C = feat.shape
        E, nq, ld = params.shape
        lib = _lib.load()
        dparams = torch.empty_like(params)
        dfeat = torch.zeros_like(feat)
        scratch = torch.empty(lib.gf_mask_head_bwd_scratch_floats(N, E * nq), dtype=torch.float32, device=feat.device)
        base = params.data_ptr()
        o_w2, o_b1 = C * (C + 3), C * (C + 3) + C
        check(lib.gf_mask_head_bwd_episodes(ptr(feat), ptr(coords), ptr(geo), ptr(qxyz), ptr(mx), base, base + 4 * o_b1,
                                            base + 4 * o_w2, ptr(gout.contiguous()), ld, N, nq, E, C, ptr(dparams),
                                            ptr(dfeat), ptr(scratch), stream_ptr()), "gf_mask_head_bwd_episodes")
        return dfeat, dparams, None, None, None, None


def mask_head_train_episodes(feat, params, coords, geo, qxyz, sqrt_max_geo):
    """logits [E,nq,N] with autograd through feat [N,16] and params [E,nq,337]."""
    for t, name in ((feat, "feat"), (coords, "coords"), (qxyz, "qxyz"), (params, "params")):
        _f32c(t, name)
    return _MaskHeadEpisodesFn.apply(feat, params, coords, geo, qxyz, sqrt_max_geo)


def mask_head_train(feat, params, coords, geo, qxyz, sqrt_max_geo):
    """logits [nq,N] with autograd through feat [N,16] and params [nq,337] (fused forward AND backward)."""
    for t, name in ((feat, "feat"), (coords, "coords"), (qxyz, "qxyz"), (params, "params")):
        _f32c(t, name)
    return _MaskHeadFn.apply(feat, params, coords, geo, qxyz, sqrt_max_geo)


class PointwiseChain:
    """Folded parameters of a Conv1d(k=1)/Linear + eval BatchNorm1d + ReLU stack for gf_pointwise_mlp."""

    def __init__(self, modules):
        import ctypes

        layers = []  # [W, scale, shift, relu]
        for m in modules:
            name = type(m).__name__
            if isinstance(m, (torch.nn.Linear, torch.nn.Conv1d, torch.nn.Conv2d)):
                W = m.weight.detach()
                W = W.reshape(W.shape[0], W.shape[1]).float().contiguous()
                bias = m.bias.detach().float() if m.bias is not None else torch.zeros(W.shape[0], device=W.device)
                layers.append([W, torch.ones_like(bias), bias.clone(), 0])
            elif isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                if m.training:
                    raise RuntimeError("PointwiseChain folds eval-mode BatchNorm only")
                s = (m.weight / torch.sqrt(m.running_var + m.eps)).detach().float()
                layers[-1][2] = (layers[-1][2] - m.running_mean.detach()) * s + m.bias.detach()
                layers[-1][1] = layers[-1][1] * s
            elif isinstance(m, torch.nn.ReLU):
                layers[-1][3] = 1
            elif isinstance(m, (torch.nn.Identity, torch.nn.Dropout)):
                continue
            else:
                raise RuntimeError(f"PointwiseChain: unsupported module {name}")
        self.tensors = [[l[0], l[1].contiguous(), l[2].contiguous()] for l in layers]
        n = len(layers)
        self.n = n
        self.channels = [layers[0][0].shape[1]] + [l[0].shape[0] for l in layers]
        mk = lambda k: (ctypes.c_void_p * n)(*[t[k].data_ptr() for t in self.tensors])  # noqa: E731
        self.W, self.scale, self.shift = mk(0), mk(1), mk(2)
        self.ch = (ctypes.c_int * (n + 1))(*self.channels)
        self.relu = (ctypes.c_int * n)(*[l[3] for l in layers])

    @staticmethod
    def supported(modules):
        ch = []
        for m in modules:
            if isinstance(m, (torch.nn.Linear, torch.nn.Conv1d)):
                if isinstance(m, torch.nn.Conv1d) and (m.kernel_size != (1,) or m.groups != 1 or m.stride != (1,)):
                    return False
                ch.append((m.weight.shape[1], m.weight.shape[0]))
        if not 1 <= len(ch) <= 4:
            return False
        ok = all(ci % 16 == 0 and 16 <= ci <= 64 for ci, _ in ch) and all(co % 16 == 0 and co <= 64 for _, co in ch[:-1])
        return ok and 1 <= ch[-1][1] <= 64


def pointwise_mlp(x, chain, rows=None):
    """Fused per-point MLP chain over the rows of x [N, C0] -> [N, C_last] (include/geoformer_hip.h); with
    rows (int32 [N]) over x[rows] without materialising the gather."""
    _f32c(x, "x")
    N = x.shape[0]
    if rows is not None:
        _i32c(rows, "rows")
        N = rows.shape[0]
    if x.shape[1] != chain.channels[0]:
        raise RuntimeError(f"pointwise_mlp: x has {x.shape[1]} channels, the chain expects {chain.channels[0]}")
    out = torch.empty((N, chain.channels[-1]), dtype=torch.float32, device=x.device)
    check(_lib.load().gf_pointwise_mlp_rows(ptr(x), ptr(rows), N, chain.n, chain.W, chain.scale, chain.shift, chain.ch,
                                            chain.relu, ptr(out), stream_ptr()), "gf_pointwise_mlp")
    return out


def group_mlp_max(grouped, chain):
    """Fused SharedMLP + max over the samples: grouped [B,C0,npoint,nsample] -> [B,C_last,npoint]."""
    _f32c(grouped, "grouped")
    B, c0, npnt, ns = grouped.shape
    if c0 != chain.channels[0]:
        raise RuntimeError(f"group_mlp_max: grouped has {c0} channels, the chain expects {chain.channels[0]}")
    out = torch.empty((B, chain.channels[-1], npnt), dtype=torch.float32, device=grouped.device)
    check(_lib.load().gf_group_mlp_max(ptr(grouped), B, npnt, ns, chain.n, chain.W, chain.scale, chain.shift, chain.ch,
                                       chain.relu, ptr(out), stream_ptr()), "gf_group_mlp_max")
    return out


def point_grid_build(xyz, radius):
    """Hash grid of one point set [1,n,3] with cells of `radius` for the grid ball query (returns the scratch tensor
    to hand to sa_group_mlp_max(grid=...)); runs on the current stream."""
    _f32c(xyz, "xyz")
    n = xyz.shape[1]
    lib = _lib.load()
    scratch = torch.empty(lib.gf_knn_scratch_bytes(n) // 4 + 16, dtype=torch.int32, device=xyz.device)
    check(lib.gf_point_grid_build(ptr(xyz), n, float(radius), ptr(scratch), stream_ptr()), "gf_point_grid_build")
    return scratch


def sa_group_mlp_max(xyz, feats, inds, radius, nsample, use_xyz, normalize_xyz, chain, grid=None):
    """Set-abstraction stage for given sample indices, fused: returns (new_xyz [B,np,3], idx [B,np,ns] int32,
    pooled [B,C_last,np]).  xyz [B,n,3], feats [B,C,n] (or None), inds int32 [B,np]."""
    _f32c(xyz, "xyz"); _i32c(inds, "inds")
    B, n, _ = xyz.shape
    C = 0
    if feats is not None:
        _f32c(feats, "feats")
        C = feats.shape[1]
    npnt = inds.shape[1]
    dev = xyz.device
    new_xyz = torch.empty((B, npnt, 3), dtype=torch.float32, device=dev)
    idx = torch.empty((B, npnt, nsample), dtype=torch.int32, device=dev)
    out = torch.empty((B, chain.channels[-1], npnt), dtype=torch.float32, device=dev)
    lib = _lib.load()
    scratch, ready = grid, grid is not None
    if scratch is None and B == 1 and n >= 4096:  # hash grid for the ball query, built here
        scratch = torch.empty(lib.gf_knn_scratch_bytes(n) // 4 + 16, dtype=torch.int32, device=dev)
    check(lib.gf_sa_group_mlp_max(ptr(xyz), ptr(feats) if feats is not None else None, ptr(inds), B, n, C, npnt,
                                  float(radius), int(nsample), int(bool(use_xyz)), int(bool(normalize_xyz)),
                                  chain.n, chain.W, chain.scale, chain.shift, chain.ch, chain.relu,
                                  ptr(new_xyz), ptr(idx), ptr(out), ptr(scratch), int(ready), stream_ptr()),
          "gf_sa_group_mlp_max")
    return new_xyz, idx, out


def decoder_stage_tables(layer, final_norm):
    """(post, pre) device-pointer tables of one decoder layer for gf_decoder_token_stage (header order)."""
    import ctypes

    sa = layer.self_attn
    post = [layer.out_mlp[0].weight, layer.out_mlp[0].bias, layer.norm3.weight, layer.norm3.bias, layer.linear1.weight,
            layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, final_norm.weight, final_norm.bias]
    pre = [layer.norm1.weight, layer.norm1.bias, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight,
           sa.out_proj.bias, layer.norm2.weight, layer.norm2.bias, layer.attn_mlp[0].weight, layer.attn_mlp[0].bias]
    for t in post + pre:
        _f32c(t.data, "decoder parameter")
    mk = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])  # noqa: E731
    return mk(post), mk(pre)


def decoder_stage_tensors(layer, final_norm):
    """(pre, post) parameter lists of one decoder layer in gf_decoder_token_stage's table order."""
    sa = layer.self_attn
    pre = [layer.norm1.weight, layer.norm1.bias, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight,
           sa.out_proj.bias, layer.norm2.weight, layer.norm2.bias, layer.attn_mlp[0].weight, layer.attn_mlp[0].bias]
    post = [layer.out_mlp[0].weight, layer.out_mlp[0].bias, layer.norm3.weight, layer.norm3.bias, layer.linear1.weight,
            layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, final_norm.weight, final_norm.bias]
    return pre, post


def _ptr_table(ts):
    import ctypes

    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def _split_grads(grads, params):
    outs, o = [], 0
    for t in params:
        outs.append(grads[o:o + t.numel()].view(t.shape))
        o += t.numel()
    assert o == grads.numel()
    return outs


class _DecoderPreTrainFn(torch.autograd.Function):
    """norm1 -> self-attention -> residual -> norm2 -> query half of the cross-attention's first linear, with the layer's
    dropouts: forward and backward in csrc/decoder_layer_train.hip.  x, qpos [B,T,64] -> (t2n, q1)."""

    @staticmethod
    def forward(ctx, x, qpos, layer, p, seed, *params):
        lib = _lib.load()
        B, T, _ = x.shape
        x, qpos = _f32c(x.contiguous(), "x"), _f32c(qpos.contiguous(), "qpos")
        params = [_f32c(t.detach().contiguous(), "decoder parameter") for t in params]
        t2n, q1 = torch.empty_like(x), torch.empty_like(x)
        save = torch.empty(lib.gf_decoder_pre_train_save_bytes(T, B) // 4, dtype=torch.float32, device=x.device)
        check(lib.gf_decoder_pre_train_fwd(ptr(x), ptr(qpos), T, B, _ptr_table(params), float(p), int(seed), int(layer),
                                           ptr(save), ptr(t2n), ptr(q1), stream_ptr()), "gf_decoder_pre_train_fwd")
        ctx.save_for_backward(x, qpos, t2n, save, *params)
        ctx.cfg = (int(layer), float(p), int(seed))
        ctx.set_materialize_grads(False)
        return t2n, q1

    @staticmethod
    def backward(ctx, d_t2n, d_q1):
        lib = _lib.load()
        x, qpos, t2n, save, *params = ctx.saved_tensors
        layer, p, seed = ctx.cfg
        B, T, _ = x.shape
        if d_t2n is None and d_q1 is None:
            return (None,) * (5 + len(params))
        d_t2n = None if d_t2n is None else _f32c(d_t2n.contiguous(), "d_t2n")
        d_q1 = None if d_q1 is None else _f32c(d_q1.contiguous(), "d_q1")
        work = torch.empty(lib.gf_decoder_pre_train_work_bytes(T, B) // 4, dtype=torch.float32, device=x.device)
        grads = torch.empty(lib.gf_decoder_pre_grad_floats(), dtype=torch.float32, device=x.device)
        dx, dqpos = torch.empty_like(x), torch.empty_like(x)
        check(lib.gf_decoder_pre_train_bwd(ptr(x), ptr(qpos), ptr(t2n), ptr(d_t2n), ptr(d_q1), T, B, _ptr_table(params), p,
                                           seed, layer, ptr(save), ptr(work), ptr(dx), ptr(dqpos), ptr(grads),
                                           stream_ptr()), "gf_decoder_pre_train_bwd")
        return (dx, dqpos, None, None, None) + tuple(_split_grads(grads, params))


class _DecoderPostTrainFn(torch.autograd.Function):
    """out_mlp -> residual with the normed query -> norm3 -> FFN -> residual -> decoder.norm, with the layer's dropouts.
    ca, t2n [B,T,64] -> (x3, inter)."""

    @staticmethod
    def forward(ctx, ca, t2n, layer, p, seed, *params):
        lib = _lib.load()
        B, T, _ = ca.shape
        ca, t2n = _f32c(ca.contiguous(), "ca"), _f32c(t2n.contiguous(), "t2n")
        params = [_f32c(t.detach().contiguous(), "decoder parameter") for t in params]
        ff = params[4].shape[0]
        x3, inter = torch.empty_like(ca), torch.empty_like(ca)
        save = torch.empty(lib.gf_decoder_post_train_save_bytes(T, B, ff) // 4, dtype=torch.float32, device=ca.device)
        check(lib.gf_decoder_post_train_fwd(ptr(ca), ptr(t2n), T, B, ff, _ptr_table(params), float(p), int(seed), int(layer),
                                            ptr(save), ptr(x3), ptr(inter), stream_ptr()), "gf_decoder_post_train_fwd")
        ctx.save_for_backward(ca, x3, save, *params)
        ctx.cfg = (int(layer), float(p), int(seed), ff)
        ctx.set_materialize_grads(False)
        return x3, inter

    @staticmethod
    def backward(ctx, d_x3, d_inter):
        lib = _lib.load()
        ca, x3, save, *params = ctx.saved_tensors
        layer, p, seed, ff = ctx.cfg
        B, T, _ = ca.shape
        if d_x3 is None and d_inter is None:
            return (None,) * (5 + len(params))
        d_x3 = None if d_x3 is None else _f32c(d_x3.contiguous(), "d_x3")
        d_inter = None if d_inter is None else _f32c(d_inter.contiguous(), "d_inter")
        work = torch.empty(lib.gf_decoder_post_train_work_bytes(T, B, ff) // 4, dtype=torch.float32, device=ca.device)
        grads = torch.empty(lib.gf_decoder_post_grad_floats(ff), dtype=torch.float32, device=ca.device)
        d_ca, d_t2n = torch.empty_like(ca), torch.empty_like(ca)
        check(lib.gf_decoder_post_train_bwd(ptr(ca), ptr(x3), ptr(d_x3), ptr(d_inter), T, B, ff, _ptr_table(params), p, seed,
                                            layer, ptr(save), ptr(work), ptr(d_ca), ptr(d_t2n), ptr(grads), stream_ptr()),
              "gf_decoder_post_train_bwd")
        return (d_ca, d_t2n, None, None, None) + tuple(_split_grads(grads, params))


def decoder_pre_train(x, qpos, layer_index, p, seed, pre_params):
    return _DecoderPreTrainFn.apply(x, qpos, layer_index, p, seed, *pre_params)


def decoder_post_train(ca, t2n, layer_index, p, seed, post_params):
    return _DecoderPostTrainFn.apply(ca, t2n, layer_index, p, seed, *post_params)


def decoder_token_stage(attn_out, tgt_in, query_pos, nq, B, nhead, ff, post, pre, state, inter_out, q1_out):
    """One fused token-side stage between two cross-attentions (include/geoformer_hip.h)."""
    check(_lib.load().gf_decoder_token_stage(ptr(attn_out), ptr(tgt_in), ptr(query_pos), nq, B, 64, nhead, ff, post,
                                             pre, ptr(state), ptr(inter_out), ptr(q1_out), stream_ptr()),
          "gf_decoder_token_stage")


def decoder_token_state(nq, B, device):
    return torch.empty(_lib.load().gf_decoder_token_state_bytes(nq, B) // 4, dtype=torch.float32, device=device)


def proposal_stats(mask_logits, cls_logits, sem_prob, logit_thresh, score_thresh, npoint_thresh, min_class=4,
                   class_major=False):
    """Fused statistics of generate_proposal: (cls_pred i32[nq], npoints i32[nq], scores f32[nq], final i32[nq]).
    sem_prob [N,ncls] is handed to the kernel class-major (one coalesced row per predicted class); pass it as
    [ncls,N] with class_major=True when the caller already has that copy."""
    _f32c(mask_logits, "mask_logits"), _f32c(cls_logits, "cls_logits")
    nq, N = mask_logits.shape
    ncls = cls_logits.shape[1]
    if sem_prob.shape != ((ncls, N) if class_major else (N, ncls)):
        raise RuntimeError(f"sem_prob must be [{N},{ncls}] ([{ncls},{N}] class-major), got {tuple(sem_prob.shape)}")
    sem_prob = _f32c(sem_prob if class_major else sem_prob.t().contiguous(), "sem_prob")
    ints = torch.empty((3, nq), dtype=torch.int32, device=mask_logits.device)
    scores = torch.empty(nq, dtype=torch.float32, device=mask_logits.device)
    check(_lib.load().gf_proposal_stats(ptr(mask_logits), ptr(cls_logits), ptr(sem_prob), nq, N, ncls,
                                        float(logit_thresh), float(score_thresh), int(npoint_thresh), int(min_class),
                                        ptr(ints[0]), ptr(ints[1]), ptr(scores), ptr(ints[2]), stream_ptr()),
          "gf_proposal_stats")
    return ints[0], ints[1], scores, ints[2]


def proposal_stats_fs(mask_logits, sim, logit_thresh, score_thresh, npoint_thresh, sim_thresh):
    """Few-shot proposal statistics (geoformer_fs.py:205-238): (npoints i32[nq], scores f32[nq], final i32[nq]) of
    the queries' mask-logit rows [nq,N] and their similarities to the support prototype [nq]."""
    _f32c(mask_logits, "mask_logits"), _f32c(sim, "sim")
    nq, N = mask_logits.shape
    if sim.shape != (nq,):
        raise RuntimeError(f"sim must be [{nq}], got {tuple(sim.shape)}")
    ints = torch.empty((2, nq), dtype=torch.int32, device=mask_logits.device)
    scores = torch.empty(nq, dtype=torch.float32, device=mask_logits.device)
    check(_lib.load().gf_proposal_stats_fs(ptr(mask_logits), ptr(sim), nq, N, float(logit_thresh), float(score_thresh),
                                           int(npoint_thresh), float(sim_thresh), ptr(ints[0]), ptr(scores),
                                           ptr(ints[1]), stream_ptr()), "gf_proposal_stats_fs")
    return ints[0], scores, ints[1]


def proposal_select(final, cls_pred, scores):
    """Accepted queries compacted on the device: (sel i32[nq], cls i64[nq], scores f32[nq], count i32[1]); the first
    `count` entries are valid."""
    _i32c(final, "final"); _i32c(cls_pred, "cls_pred"); _f32c(scores, "scores")
    nq = final.shape[0]
    dev = final.device
    sel = torch.empty(nq, dtype=torch.int32, device=dev)
    cls = torch.empty(nq, dtype=torch.int64, device=dev)
    sc = torch.empty(nq, dtype=torch.float32, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    check(_lib.load().gf_proposal_select(ptr(final), ptr(cls_pred), ptr(scores), nq, ptr(sel), ptr(cls), ptr(sc), ptr(cnt),
                                         stream_ptr()), "gf_proposal_select")
    return sel, cls, sc, cnt


def proposal_scatter(mask_logits, sel, fg_idxs, logit_thresh, num_points):
    """0/1 membership rows [len(sel), num_points] (int32) of the selected queries over the scene's points."""
    _f32c(mask_logits, "mask_logits"), _i32c(sel, "sel")
    if not (fg_idxs.is_cuda and fg_idxs.dtype == torch.int64 and fg_idxs.is_contiguous()):
        raise RuntimeError("fg_idxs: expected a contiguous int64 tensor on the GPU")
    out = torch.zeros((sel.shape[0], num_points), dtype=torch.int32, device=mask_logits.device)
    check(_lib.load().gf_proposal_scatter(ptr(mask_logits), ptr(sel), sel.shape[0], mask_logits.shape[1],
                                          ptr(fg_idxs), float(logit_thresh), int(num_points), ptr(out), stream_ptr()),
          "gf_proposal_scatter")
    return out


def relpos_prepare(geo, inds):
    """(geo_ctx [nq,nc], max_geo [nq]) of one scene: geo[:, inds] and its row maxima with all-unreachable rows set to
    the largest maximum (ingredients of the relative position embedding, csrc/proposal.hip)."""
    _f32c(geo, "geo"); _i32c(inds, "inds")
    nq, n = geo.shape
    nc = inds.shape[0]
    geo_ctx = torch.empty((nq, nc), dtype=torch.float32, device=geo.device)
    max_geo = torch.empty(nq, dtype=torch.float32, device=geo.device)
    check(_lib.load().gf_relpos_prepare(ptr(geo), ptr(inds), nq, n, nc, ptr(geo_ctx), ptr(max_geo), stream_ptr()),
          "gf_relpos_prepare")
    return geo_ctx, max_geo


def mask_intersections(masks):
    """inter[i,j] = |mask_i AND mask_j| for 0/1 int32 masks [n,N] (bit-packed popcount kernel)."""
    _i32c(masks, "masks")
    n, N = masks.shape
    lib = _lib.load()
    scratch = torch.empty(lib.gf_mask_intersections_scratch_bytes(n, N) // 8 + 1, dtype=torch.int64,
                          device=masks.device)
    inter = torch.empty((n, n), dtype=torch.int32, device=masks.device)
    check(lib.gf_mask_intersections(ptr(masks), n, N, ptr(scratch), ptr(inter), stream_ptr()), "gf_mask_intersections")
    return inter


def backbone_transformer_params(before, transformer, after):
    """Device-pointer table of gf_backbone_transformer in the order include/geoformer_hip.h documents."""
    import ctypes

    ts = [before.weight, before.bias, transformer.position_linear.weight, transformer.position_linear.bias]
    for layer in transformer.layers:
        a, ff = layer.attn_1, layer.ff
        ts += [layer.norm_1.alpha, layer.norm_1.bias, a.q_linear.weight, a.q_linear.bias, a.k_linear.weight,
               a.k_linear.bias, a.v_linear.weight, a.v_linear.bias, a.out.weight, a.out.bias, layer.norm_2.alpha,
               layer.norm_2.bias, ff.linear_1.weight, ff.linear_1.bias, ff.linear_2.weight, ff.linear_2.bias]
    ts += [transformer.norm.alpha, transformer.norm.bias, after.weight, after.bias]
    for t in ts:
        _f32c(t.data, "transformer parameter")
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts]), len(transformer.layers)


def backbone_transformer(feats, coords, scene_offsets, n_scenes, params, n_layers):
    """Fused before-linear -> per-scene voxel transformer -> after-linear: feats [M,c], coords int32 [M,4],
    scene_offsets int32 [n_scenes+1] on the device.  Returns [M,c]."""
    _f32c(feats, "feats")
    if coords.dtype != torch.int32 or not coords.is_contiguous():
        raise RuntimeError("coords must be contiguous int32 [M,4]")
    lib = _lib.load()
    M, c = feats.shape
    out = torch.empty_like(feats)
    scratch = torch.empty(lib.gf_backbone_transformer_scratch_bytes(M) // 4, dtype=torch.float32, device=feats.device)
    check(lib.gf_backbone_transformer(ptr(feats), ptr(coords), ptr(scene_offsets), n_scenes, M, c, n_layers, params,
                                      ptr(scratch), ptr(out), stream_ptr()), "gf_backbone_transformer")
    return out


def backbone_transformer_tensors(before, transformer, after):
    """The parameters of the voxel transformer stack in the order of gf_backbone_transformer's table."""
    ts = [before.weight, before.bias, transformer.position_linear.weight, transformer.position_linear.bias]
    for layer in transformer.layers:
        a, ff = layer.attn_1, layer.ff
        ts += [layer.norm_1.alpha, layer.norm_1.bias, a.q_linear.weight, a.q_linear.bias, a.k_linear.weight,
               a.k_linear.bias, a.v_linear.weight, a.v_linear.bias, a.out.weight, a.out.bias, layer.norm_2.alpha,
               layer.norm_2.bias, ff.linear_1.weight, ff.linear_1.bias, ff.linear_2.weight, ff.linear_2.bias]
    return ts + [transformer.norm.alpha, transformer.norm.bias, after.weight, after.bias]


def dropout_keep_reference(seed, p, site, rows, cols):
    """The keep factor (0 or 1/(1-p)) gf_backbone_transformer_train_* uses for element (row, col) of a dropout site, as
    the header states it -- for tests that rebuild the masks.  rows, cols: integer tensors (broadcast together)."""
    M32 = 0xFFFFFFFF

    def fmix(x):
        x = x ^ (x >> 16)
        x = (x * 0x85EBCA6B) & M32
        x = x ^ (x >> 13)
        x = (x * 0xC2B2AE35) & M32
        return x ^ (x >> 16)

    rows, cols = rows.to(torch.int64), cols.to(torch.int64)
    h = fmix((int(seed) & M32) ^ ((rows * 64 + int(site)) & M32))
    h = fmix((h + ((cols * 0x9E3779B1) & M32)) & M32)
    if p <= 0:
        return torch.ones_like(h, dtype=torch.float32)
    keep = (h >> 8) >= int(float(p) * 16777216.0)
    return keep.to(torch.float32) * float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))


class _VoxelTransformerTrainFn(torch.autograd.Function):
    """before-linear -> per-scene voxel transformer (with its dropouts) -> after-linear of a deep U-Net level, forward and
    backward in csrc/backbone_attn.hip (gf_backbone_transformer_train_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, feats, coords, n_scenes, n_layers, p, seed, *params):
        import ctypes

        lib = _lib.load()
        M, c = feats.shape
        feats = _f32c(feats.contiguous(), "feats")
        params = [_f32c(t.detach().contiguous(), "transformer parameter") for t in params]
        table = (ctypes.c_void_p * len(params))(*[t.data_ptr() for t in params])
        out = torch.empty_like(feats)
        save = torch.empty(lib.gf_backbone_transformer_train_save_bytes(M, n_layers) // 4, dtype=torch.float32,
                           device=feats.device)
        check(lib.gf_backbone_transformer_train_fwd(ptr(feats), ptr(coords), n_scenes, M, c, n_layers, table, float(p),
                                                    int(seed), ptr(save), ptr(out), stream_ptr()),
              "gf_backbone_transformer_train_fwd")
        ctx.save_for_backward(feats, save, *params)
        ctx.cfg = (n_scenes, n_layers, float(p), int(seed))
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes

        lib = _lib.load()
        feats, save, *params = ctx.saved_tensors
        n_scenes, n_layers, p, seed = ctx.cfg
        M, c = feats.shape
        dout = _f32c(dout.contiguous(), "dout")
        table = (ctypes.c_void_p * len(params))(*[t.data_ptr() for t in params])
        work = torch.empty(lib.gf_backbone_transformer_train_work_bytes(M, n_layers, n_scenes) // 4, dtype=torch.float32,
                           device=feats.device)
        grads = torch.empty(lib.gf_backbone_transformer_grad_floats(c, n_layers), dtype=torch.float32, device=feats.device)
        dfeats = torch.empty_like(feats)
        check(lib.gf_backbone_transformer_train_bwd(ptr(feats), ptr(dout), n_scenes, M, c, n_layers, table, p, seed,
                                                    ptr(save), ptr(work), ptr(dfeats), ptr(grads), stream_ptr()),
              "gf_backbone_transformer_train_bwd")
        outs, o = [], 0
        for t in params:
            outs.append(grads[o:o + t.numel()].view(t.shape))
            o += t.numel()
        assert o == grads.numel()
        return (dfeats, None, None, None, None, None) + tuple(outs)


def backbone_transformer_train_supported(feats, coords, transformer):
    return (feats.is_cuda and feats.dtype == torch.float32 and feats.shape[0] > 0 and feats.shape[1] % 16 == 0
            and feats.shape[1] <= 384 and coords.dtype == torch.int32 and coords.is_contiguous()
            and transformer.d_model == 128 and 1 <= len(transformer.layers) <= 4
            and len({float(m.p) for m in transformer.modules() if isinstance(m, torch.nn.Dropout)}) == 1
            and all(l.attn_1.h == 4 and l.ff.linear_1.out_features == 64 for l in transformer.layers)
            and os.environ.get("GF_FUSED_VOXEL_TRANSFORMER", "1") != "0")


def backbone_transformer_train(feats, coords, n_scenes, before, transformer, after, seed=None):
    """Training forward of a deep level's voxel transformer stack with a native backward.  The dropouts are active when
    the transformer module is in training mode (p of its modules); `seed` defaults to a draw from the framework's CPU
    generator, so torch.manual_seed fixes the masks."""
    layer0 = transformer.layers[0]
    p = float(layer0.dropout_1.p) if transformer.training else 0.0
    if seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    ts = backbone_transformer_tensors(before, transformer, after)
    return _VoxelTransformerTrainFn.apply(feats, coords, int(n_scenes), len(transformer.layers), p, seed, *ts)


def decoder_pack_weights(W1, W2, Wv):
    lib = _lib.load()
    wp = torch.empty(lib.gf_decoder_wpack_floats(), dtype=torch.float32, device=W1.device)
    check(lib.gf_decoder_pack_weights(ptr(_f32c(W1, "W1")), ptr(_f32c(W2, "W2")), ptr(_f32c(Wv, "Wv")), ptr(wp),
                                      stream_ptr()), "gf_decoder_pack_weights")
    return wp


_launch_cfg = threading.local()


@contextlib.contextmanager
def co_resident_launches():
    """Inside this context the inference cross-attention is launched in its 8-wave workgroup shape
    (gf_decoder_cross_attn_cfg), which fits on a compute unit beside a BFS workgroup: SplitForward.finish() queues one
    scene's decoder under the next scene's sampling / BFS stretch.  Per host thread."""
    old = getattr(_launch_cfg, "cross_attn_waves", 16)
    _launch_cfg.cross_attn_waves = 8
    try:
        yield
    finally:
        _launch_cfg.cross_attn_waves = old


def co_resident():
    """Whether this host thread is inside ``co_resident_launches`` (a serving loop's SplitForward.finish())."""
    return getattr(_launch_cfg, "cross_attn_waves", 16) != 16


def decoder_cross_attn(geo_ctx, max_geo, qloc, cloc, lo, hi, gaussB, Q1, K1, Kv, wpack, b2):
    """Fused vector cross-attention; shapes as in include/geoformer_hip.h.  Returns [B,nq,64]."""
    for t, name in ((geo_ctx, "geo_ctx"), (max_geo, "max_geo"), (qloc, "qloc"), (cloc, "cloc"), (lo, "lo"), (hi, "hi"),
                    (gaussB, "gaussB"), (Q1, "Q1"), (K1, "K1"), (Kv, "Kv"), (wpack, "wpack"), (b2, "b2")):
        _f32c(t, name)
    B, nq, nc = geo_ctx.shape
    d = Q1.shape[-1]
    out = torch.empty((B, nq, d), dtype=torch.float32, device=Q1.device)
    check(_lib.load().gf_decoder_cross_attn_cfg(ptr(geo_ctx), ptr(max_geo), ptr(qloc), ptr(cloc), ptr(lo), ptr(hi),
                                                ptr(gaussB), ptr(Q1), ptr(K1), ptr(Kv), ptr(wpack), ptr(b2), B, nq, nc, d,
                                                ptr(out), None, None, getattr(_launch_cfg, "cross_attn_waves", 16),
                                                stream_ptr()), "gf_decoder_cross_attn")
    return out


class _CrossAttnFn(torch.autograd.Function):
    """Fused vector cross-attention with a fused, recompute-based backward (csrc/decoder_attn.hip).  Differentiable
    inputs: Q1 [B,nq,64], K1, Kv [B,nc,64] (the hoisted projections) and the pair weights W1, W2, Wv [64,64]; the
    geodesic embedding inputs are data."""

    @staticmethod
    def forward(ctx, geo_ctx, max_geo, qloc, cloc, lo, hi, gaussB, Q1, K1, Kv, W1, W2, Wv):
        lib = _lib.load()
        B, nq, nc = geo_ctx.shape
        d = Q1.shape[-1]
        wpack = decoder_pack_weights(W1.detach().contiguous(), W2.detach().contiguous(), Wv.detach().contiguous())
        out = torch.empty((B, nq, d), dtype=torch.float32, device=Q1.device)
        sm, sl = torch.empty_like(out), torch.empty_like(out)
        Q1, K1, Kv = Q1.contiguous(), K1.contiguous(), Kv.contiguous()
        check(lib.gf_decoder_cross_attn(ptr(geo_ctx), ptr(max_geo), ptr(qloc), ptr(cloc), ptr(lo), ptr(hi), ptr(gaussB),
                                        ptr(Q1), ptr(K1), ptr(Kv), ptr(wpack), None, B, nq, nc, d, ptr(out), ptr(sm),
                                        ptr(sl), stream_ptr()), "gf_decoder_cross_attn")
        ctx.save_for_backward(geo_ctx, max_geo, qloc, cloc, lo, hi, gaussB, Q1, K1, Kv, wpack, W2.detach().contiguous(),
                              out, sm, sl)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        geo_ctx, max_geo, qloc, cloc, lo, hi, gaussB, Q1, K1, Kv, wpack, W2, out, sm, sl = ctx.saved_tensors
        B, nq, nc = geo_ctx.shape
        d = Q1.shape[-1]
        dQ1 = torch.empty_like(Q1)
        dK1, dKv = torch.zeros_like(K1), torch.zeros_like(Kv)
        dW = torch.empty((3, d, d), dtype=torch.float32, device=Q1.device)
        scratch = torch.empty(lib.gf_decoder_cross_attn_bwd_scratch_floats(B, nq, nc), dtype=torch.float32,
                              device=Q1.device)
        check(lib.gf_decoder_cross_attn_bwd(ptr(geo_ctx), ptr(max_geo), ptr(qloc), ptr(cloc), ptr(lo), ptr(hi),
                                            ptr(gaussB), ptr(Q1), ptr(K1), ptr(Kv), ptr(wpack), ptr(W2), ptr(out),
                                            ptr(sm), ptr(sl), ptr(gout.contiguous()), B, nq, nc, d, ptr(dQ1), ptr(dK1),
                                            ptr(dKv), ptr(dW), ptr(scratch), stream_ptr()),
              "gf_decoder_cross_attn_bwd")
        return None, None, None, None, None, None, None, dQ1, dK1, dKv, dW[0], dW[1], dW[2]


def decoder_cross_attn_train(geo_ctx, max_geo, qloc, cloc, lo, hi, gaussB, Q1, K1, Kv, W1, W2, Wv):
    """[B,nq,64] with autograd through Q1, K1, Kv, W1, W2, Wv (fused forward AND backward)."""
    for t, name in ((geo_ctx, "geo_ctx"), (max_geo, "max_geo"), (qloc, "qloc"), (cloc, "cloc"), (lo, "lo"), (hi, "hi"),
                    (gaussB, "gaussB")):
        _f32c(t, name)
    return _CrossAttnFn.apply(geo_ctx, max_geo, qloc, cloc, lo, hi, gaussB, Q1, K1, Kv, W1, W2, Wv)


class _SoftmaxDim1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        x = x.contiguous()
        n0, n1 = x.shape[0], x.shape[1]
        inner = x.numel() // max(n0 * n1, 1)
        y = torch.empty_like(x)
        check(_lib.load().gf_softmax_dim1_fwd(ptr(x), n0, n1, inner, float(scale), ptr(y), stream_ptr()),
              "gf_softmax_dim1_fwd")
        ctx.save_for_backward(y)
        ctx.scale = float(scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gy = gy.contiguous()
        n0, n1 = y.shape[0], y.shape[1]
        inner = y.numel() // max(n0 * n1, 1)
        gx = torch.empty_like(y)
        check(_lib.load().gf_softmax_dim1_bwd(ptr(y), ptr(gy), n0, n1, inner, ctx.scale, ptr(gx), stream_ptr()),
              "gf_softmax_dim1_bwd")
        return gx, None


def softmax_dim1(x, scale=1.0):
    """softmax(scale * x, dim=1) for a float32 GPU tensor with >= 2 dims, differentiable (streaming HIP kernels)."""
    _f32c(x.detach() if x.is_contiguous() else x.detach().contiguous(), "x")
    return _SoftmaxDim1.apply(x, scale)


# ---- training-mode BatchNorm1d + ReLU over voxel rows (csrc/bn_train.hip) -----------------------------------------
class _BNReLUTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, relu):
        lib = _lib.load()
        M, C = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((2, C), dtype=torch.float32, device=x.device)
        scratch = torch.empty(lib.gf_bn_train_scratch_floats(M, C), dtype=torch.float32, device=x.device)
        check(lib.gf_bn_relu_train_fwd(ptr(x), M, C, ptr(weight), ptr(bias), float(eps), float(momentum), int(relu),
                                       ptr(running_mean), ptr(running_var), ptr(y), stats[0].data_ptr(),
                                       stats[1].data_ptr(), ptr(scratch), stream_ptr()),
              "gf_bn_relu_train_fwd")
        ctx.save_for_backward(x, y, weight, stats)
        ctx.relu = int(relu)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, weight, stats = ctx.saved_tensors
        lib = _lib.load()
        M, C = x.shape
        gy = gy.contiguous()
        need_x = ctx.needs_input_grad[0]
        dx = torch.empty_like(x) if need_x else None
        dwb = torch.empty((2, C), dtype=torch.float32, device=x.device)
        scratch = torch.empty(lib.gf_bn_train_scratch_floats(M, C), dtype=torch.float32, device=x.device)
        check(lib.gf_bn_relu_train_bwd(ptr(x), ptr(y), ptr(gy), M, C, ptr(weight), stats[0].data_ptr(),
                                       stats[1].data_ptr(), ctx.relu, ptr(dx), dwb[0].data_ptr(), dwb[1].data_ptr(),
                                       ptr(scratch), stream_ptr()), "gf_bn_relu_train_bwd")
        return dx, dwb[0], dwb[1], None, None, None, None, None


def bn_relu_train_supported(bn, x):
    """Training-mode BatchNorm1d over fp32 voxel rows [M, C] on the GPU that the fused pair can take."""
    return (bn.training and not getattr(bn, "sync_across_ranks", False) and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= 2
            and x.shape[1] % 4 == 0 and x.shape[1] <= 256 and bn.affine and bn.track_running_stats
            and bn.momentum is not None)


def bn_relu_train(bn, x, relu=True):
    """relu(bn(x)) for a training-mode nn.BatchNorm1d `bn` over rows x [M, C]: batch statistics, running statistics
    updated like nn.BatchNorm1d does, three launches forward and three backward (autograd through x, weight, bias)."""
    _f32c(x, "x")
    y = _BNReLUTrainFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, relu)
    if hasattr(bn, "_flush_counter"):
        bn._nbt_pending = getattr(bn, "_nbt_pending", 0) + 1  # the lean subclass counts on the host (model/layers.py)
    elif bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return y


class _BNTrainCLFn(torch.autograd.Function):
    """Training-mode batch norm over the channel-major layouts [B, C, L] / [B, C, H, W] (csrc/bn_train.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, relu):
        lib = _lib.load()
        B, C = x.shape[0], x.shape[1]
        L = x.numel() // (B * C)
        y = torch.empty_like(x)
        stats = torch.empty((2, C), dtype=torch.float32, device=x.device)
        scratch = torch.empty(lib.gf_bn_train_cl_scratch_floats(B, C, L), dtype=torch.float32, device=x.device)
        check(lib.gf_bn_relu_train_cl_fwd(ptr(x), B, C, L, ptr(weight), ptr(bias), float(eps), float(momentum), int(relu),
                                          ptr(running_mean), ptr(running_var), ptr(y), stats[0].data_ptr(),
                                          stats[1].data_ptr(), ptr(scratch), stream_ptr()),
              "gf_bn_relu_train_cl_fwd")
        ctx.save_for_backward(x, y if relu else None, weight, stats)
        ctx.relu, ctx.dims = int(relu), (B, C, L)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, weight, stats = ctx.saved_tensors
        lib = _lib.load()
        B, C, L = ctx.dims
        gy = gy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dwb = torch.empty((2, C), dtype=torch.float32, device=x.device)
        scratch = torch.empty(lib.gf_bn_train_cl_scratch_floats(B, C, L), dtype=torch.float32, device=x.device)
        check(lib.gf_bn_relu_train_cl_bwd(ptr(x), ptr(y), ptr(gy), B, C, L, ptr(weight), stats[0].data_ptr(),
                                          stats[1].data_ptr(), ctx.relu, ptr(dx), dwb[0].data_ptr(), dwb[1].data_ptr(),
                                          ptr(scratch), stream_ptr()),
              "gf_bn_relu_train_cl_bwd")
        return dx, dwb[0], dwb[1], None, None, None, None, None


def bn_train_cl_supported(bn, x):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3 and x.is_contiguous() and bn.affine
            and bn.track_running_stats and bn.momentum is not None and x.shape[1] <= 256
            and x.shape[0] * x.shape[1] < 65536 and x.numel() // x.shape[1] >= 2)


def bn_train_cl(bn, x, relu=False):
    """bn(x) (optionally followed by ReLU) for a training-mode BatchNorm over x [B, C, L...]: batch statistics, running
    statistics updated; three launches forward, three backward."""
    return _BNTrainCLFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, relu)


# ---- row gathers with cheap exact gradients ---------------------------------------------------------------------------
class _TakeRowsUniqueFn(torch.autograd.Function):
    """x[idx] for an index WITHOUT repeats (the foreground list, the host-RNG draws without replacement): the gradient
    is a plain scatter of rows into zeros.  The framework's index backward does not know the rows are unique and goes
    through its accumulating index_put (sort / atomics: 0.3 ms for the [550k, 16] feature rows of a training batch)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return x[idx]

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        gx = g.new_zeros(ctx.shape)
        gx.index_copy_(0, idx, g)
        return gx, None


def take_rows_unique(x, idx):
    """x[idx] along dim 0; idx int64 without repeats (the caller's guarantee)."""
    if x.requires_grad and torch.is_grad_enabled() and x.is_cuda:
        return _TakeRowsUniqueFn.apply(x, idx)
    return x[idx]


class _PointsFromVoxelsFn(torch.autograd.Function):
    """feats[p2v_map] (voxel rows -> point rows, geoformer.py:541); the gradient is the per-voxel SUM of the points'
    rows -- the native voxel reduction over the rule table v2p_map (gf_voxelize_fp, sum mode: ascending point order,
    deterministic) instead of an accumulating index_put with atomics."""

    @staticmethod
    def forward(ctx, feats, p2v, v2p):
        ctx.save_for_backward(v2p)
        return feats[p2v.long()]

    @staticmethod
    def backward(ctx, g):
        (v2p,) = ctx.saved_tensors
        return voxelize_fp(g.contiguous(), v2p, mode=3), None, None


def points_from_voxels(feats, p2v, v2p=None):
    if (v2p is not None and feats.requires_grad and torch.is_grad_enabled() and feats.is_cuda and feats.dtype == torch.float32
            and v2p.dtype == torch.int32 and v2p.is_contiguous() and v2p.shape[0] == feats.shape[0]):
        return _PointsFromVoxelsFn.apply(feats, p2v, v2p)
    return feats[p2v.long()]
