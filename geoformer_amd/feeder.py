"""Scene feeder: collate -> pinned staging -> H2D on a copy stream -> voxelisation on the GPU, one batch ahead of the
consumer.

The reference builds every batch on the host -- ``datasets/scannetv2_inst.py:389-455`` (trainMerge / testMerge) incl.
``pointgroup_ops.voxelization_idx`` on the CPU -- and its drivers then move the tensors with blocking ``.cuda()``
calls (train.py:63-75, test.py:56).  At the rate the MI355X forward consumes scenes (5-6 ms each) that host
voxelisation (~8 ms per 150k-point scene even vectorised) and the blocking copies are the bottleneck of the loop.

Here batch i+1 is started when batch i is handed over.  The consumer's thread only allocates the batch's device
tensors and describes the job; the bytes are moved by ONE native worker thread of the library (csrc/feeder.hip: memcpy
into pinned buffers, asynchronous uploads on a copy stream, first half of ``gf_voxelize_idx`` behind them, its two
sizes on their way to a pinned word).  Rounds 2-5 did the staging memcpy on the consumer's own thread -- 0.8 ms per
150k-point scene in which that thread launches nothing -- because a Python helper thread halved the loop's rate (the
forward is bound by the host's launch rate and a second Python thread takes the interpreter lock away from it for
milliseconds at a time); a native thread holds no interpreter lock.
  start(i+1): device tensors allocated, job handed to the worker: returns at once;
  finish(i+1): (at the next hand-over) the worker has long issued everything; the sizes are there, the second half
               (maps) is queued, an event marks the batch.
The consumer's stream waits for that event only.  No device-wide synchronisation, no blocking copy.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import check, ptr

_STAGED = ("locs", "locs_float", "feats", "labels", "instance_labels", "offsets", "pc_mins", "pc_maxs")


class _InFlight:
    __slots__ = ("out", "scratch", "input_map", "head", "head_host", "voxelise", "N", "ncol", "slot", "keep")


class DeviceFeeder:
    """for batch in DeviceFeeder(raw_batches, device): model(batch, epoch, training=False)

    raw_batches: iterable of host batch dicts in ``scene.collate_raw`` layout (a dict that already carries
    ``voxel_locs`` is only uploaded)."""

    def __init__(self, raw_batches, device, mode: int = 4, reserve_points=None):
        """reserve_points: the largest batch (points) the loop will see.  Pinned staging buffers are sized for it at
        their first allocation, for all three slots: growing one later is a `pin_memory()` of tens of milliseconds in the
        middle of the loop (bench.py's test_py_shape leg: one 45-55 ms step in sixteen)."""
        self.reserve_points = int(reserve_points) if reserve_points else 0
        self.src = iter(raw_batches)
        dev = torch.device(device)
        self.device = dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())
        self.mode = mode
        self.lib = _lib.load()
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.pinned = {}  # (key, slot) -> pinned staging tensor, grown on demand
        self.slot = 0
        self.busy = set()  # slots whose job's uploads may still read the slot's staging buffers
        self.handle = self.lib.gf_feeder_create(self.device.index)
        if not self.handle:
            raise _lib.GeoFormerHipError("gf_feeder_create: " + (self.lib.gf_last_error() or b"").decode())
        self.next = self._start()

    def close(self):
        """Stop the worker thread (after the last batch; also on garbage collection)."""
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.gf_feeder_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stage(self, key, t):
        buf = self.pinned.get((key, self.slot))
        n = t.numel()
        if buf is None or buf.dtype != t.dtype or buf.numel() < n:
            # per-point tensors are sized for the reserve (or 1.3 x this batch), for every slot at once
            rows = t.shape[0] if t.dim() >= 1 else 1
            per_row = n // max(rows, 1)
            cap = max(n, 1)
            if rows > 1024:
                cap = max(int(1.3 * n), per_row * self.reserve_points)
            for sl in range(3):
                old = self.pinned.get((key, sl))
                if old is None or old.dtype != t.dtype or old.numel() < cap:
                    self.pinned[(key, sl)] = torch.empty(cap, dtype=t.dtype).pin_memory()
            buf = self.pinned[(key, self.slot)]
        return buf[:n].view(t.shape)  # (filled by the worker thread: csrc/feeder.hip)

    def _start(self):
        try:
            raw = next(self.src)
        except StopIteration:
            return None
        f = _InFlight()
        f.slot = self.slot
        if f.slot in self.busy:  # three hand-overs ago: long done
            check(self.lib.gf_feeder_wait_copied(self.handle, f.slot), "gf_feeder_wait_copied")
            self.busy.discard(f.slot)
        job = _lib.FeederJob()
        job.slot = f.slot
        job.stream = self.copy_stream.cuda_stream
        f.keep = []  # host tensors the worker reads: alive until the job has been issued
        nc = 0
        with torch.cuda.stream(self.copy_stream):
            out = {}
            for k, v in raw.items():
                if torch.is_tensor(v) and k in _STAGED:
                    v = v.contiguous()
                    if nc >= _lib.FEEDER_MAX_COPIES:
                        raise RuntimeError("DeviceFeeder: more staged tensors than GF_FEEDER_MAX_COPIES")
                    pin = self._stage(k, v)
                    dev_t = torch.empty(v.shape, dtype=v.dtype, device=self.device)
                    job.src[nc], job.pinned[nc], job.dev[nc] = v.data_ptr(), pin.data_ptr(), dev_t.data_ptr()
                    job.bytes[nc] = v.numel() * v.element_size()
                    nc += 1
                    f.keep.append(v)
                    out[k] = dev_t
                elif torch.is_tensor(v):
                    out[k] = v.to(self.device, non_blocking=True)
                else:
                    out[k] = v
            job.n_copies = nc
            f.out = out
            f.voxelise = "voxel_locs" not in out
            if f.voxelise:
                coords = out["locs"]
                if coords.dtype != torch.int64 or coords.dim() != 2:
                    raise RuntimeError("locs: expected an int64 [N,4] tensor")
                f.N, f.ncol = coords.shape
                f.scratch = torch.empty(self.lib.gf_voxelize_idx_scratch_bytes(f.N) // 8 + 1, dtype=torch.int64,
                                        device=self.device)
                f.input_map = torch.empty(f.N, dtype=torch.int32, device=self.device)
                f.head = torch.empty(3, dtype=torch.int32, device=self.device)
                f.head_host = self.pinned.get(("head", f.slot))
                if f.head_host is None:
                    f.head_host = self.pinned[("head", f.slot)] = torch.zeros(3, dtype=torch.int32).pin_memory()
                job.coords_dev, job.N, job.ncol, job.mode = coords.data_ptr(), f.N, f.ncol, int(self.mode)
                job.scratch, job.input_map = f.scratch.data_ptr(), f.input_map.data_ptr()
                job.head_dev, job.head_host = f.head.data_ptr(), f.head_host.data_ptr()
        check(self.lib.gf_feeder_submit(self.handle, ctypes.byref(job)), "gf_feeder_submit")
        self.busy.add(f.slot)
        self.slot = (self.slot + 1) % 3
        return f

    def _finish(self, f):
        out = f.out
        # the worker has queued the job's copies and kernels on the copy stream (normally long ago); only now may this
        # thread queue anything behind them there
        check(self.lib.gf_feeder_wait_issued(self.handle, f.slot), "gf_feeder (worker thread)")
        f.keep = None
        with torch.cuda.stream(self.copy_stream):
            if f.voxelise:
                check(self.lib.gf_feeder_wait_head(self.handle, f.slot), "gf_feeder_wait_head")
                M, max_active, err = f.head_host.tolist()
                if err:
                    raise _lib.GeoFormerHipError("gf_voxelize_idx: a coordinate lies outside [0, 65535] "
                                                 "(packed 16-bit key fields)")
                max_active = max(max_active, 1)
                out_coords = torch.empty((M, f.ncol), dtype=torch.int64, device=self.device)
                out_map = torch.empty((M, max_active + 1), dtype=torch.int32, device=self.device)
                check(self.lib.gf_voxelize_idx_fill(ptr(out["locs"]), f.N, f.ncol, int(self.mode), ptr(f.scratch),
                                                    ptr(f.input_map), M, max_active, ptr(out_coords), ptr(out_map),
                                                    self.copy_stream.cuda_stream), "gf_voxelize_idx_fill")
                out["voxel_locs"], out["p2v_map"], out["v2p_map"] = out_coords, f.input_map, out_map
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ready)
        for v in out.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)  # allocated on the copy stream, used on the consumer's
        # what the batch's tensors wait for: the model starts the head of the backbone (voxel features, rulebooks) behind THIS
        # event on the executor's side stream instead of behind everything the consumer's stream still has queued
        # (GeoFormer._inputs_ahead)
        from . import unet_exec
        side = unet_exec.side_stream_for(self.device, cur)
        for v in out.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(side)
        out["inputs_event"] = (ready,)
        return out

    def __iter__(self):
        return self

    def __next__(self):
        if self.next is None:
            self.close()
            raise StopIteration
        out = self._finish(self.next)
        self.next = self._start()  # the device works on it beside the scene just handed over
        return out
