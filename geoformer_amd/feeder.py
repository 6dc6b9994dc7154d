"""Scene feeder: collate -> pinned staging -> H2D on a copy stream -> voxelisation on the GPU, one batch ahead of the
consumer.

The reference builds every batch on the host -- ``datasets/scannetv2_inst.py:389-455`` (trainMerge / testMerge) incl.
``pointgroup_ops.voxelization_idx`` on the CPU -- and its drivers then move the tensors with blocking ``.cuda()``
calls (train.py:63-75, test.py:56).  At the rate the MI355X forward consumes scenes (5-6 ms each) that host
voxelisation (~8 ms per 150k-point scene even vectorised) and the blocking copies are the bottleneck of the loop.

Here batch i+1 is started when batch i is handed over, in the consumer's own thread (a helper thread was tried first
and halved the loop's rate: the forward is bound by the host's launch rate and a second Python thread takes the
interpreter lock away from it for milliseconds at a time):
  start(i+1): staged in pinned buffers, uploaded on a copy stream, first half of ``gf_voxelize_idx`` (voxel ids,
              counts) queued behind the copies, the two sizes it produces on their way to a pinned word -- all
              asynchronous, the device works on it beside scene i;
  finish(i+1): (at the next hand-over) the sizes are there, the second half (maps) is queued, an event marks the batch.
The consumer's stream waits for that event only.  No device-wide synchronisation, no blocking copy.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, ptr

_STAGED = ("locs", "locs_float", "feats", "labels", "instance_labels", "offsets", "pc_mins", "pc_maxs")


class _InFlight:
    __slots__ = ("out", "scratch", "input_map", "head_host", "head_ready", "copied", "N", "ncol")


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
        self.busy = {}  # slot -> event after which its staging buffers may be rewritten
        self.next = self._start()

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
        view = buf[:n].view(t.shape)
        # numpy's memcpy, not Tensor.copy_: a CPU-side torch copy wakes torch's intra-op thread pool, whose spinning
        # workers slow the launching thread several times over on a many-core host
        import numpy as np

        np.copyto(view.numpy(), t.numpy())
        return view

    def _start(self):
        try:
            raw = next(self.src)
        except StopIteration:
            return None
        f = _InFlight()
        ev = self.busy.get(self.slot)
        if ev is not None:
            ev.synchronize()  # two hand-overs ago: long done
        with torch.cuda.stream(self.copy_stream):
            out = {}
            for k, v in raw.items():
                if torch.is_tensor(v) and k in _STAGED:
                    out[k] = self._stage(k, v.contiguous()).to(self.device, non_blocking=True)
                elif torch.is_tensor(v):
                    out[k] = v.to(self.device, non_blocking=True)
                else:
                    out[k] = v
            f.out = out
            f.copied = torch.cuda.Event()
            f.copied.record(self.copy_stream)
            self.busy[self.slot] = f.copied
            f.head_ready = None
            if "voxel_locs" not in out:
                coords = out["locs"]
                if coords.dtype != torch.int64 or coords.dim() != 2:
                    raise RuntimeError("locs: expected an int64 [N,4] tensor")
                f.N, f.ncol = coords.shape
                f.scratch = torch.empty(self.lib.gf_voxelize_idx_scratch_bytes(f.N) // 8 + 1, dtype=torch.int64,
                                        device=self.device)
                f.input_map = torch.empty(f.N, dtype=torch.int32, device=self.device)
                head = torch.empty(3, dtype=torch.int32, device=self.device)
                check(self.lib.gf_voxelize_idx_count(ptr(coords), f.N, f.ncol, int(self.mode), ptr(f.scratch),
                                                     ptr(f.input_map), ptr(head), self.copy_stream.cuda_stream),
                      "gf_voxelize_idx_count")
                f.head_host = self.pinned.get(("head", self.slot))
                if f.head_host is None:
                    f.head_host = self.pinned[("head", self.slot)] = torch.zeros(3, dtype=torch.int32).pin_memory()
                f.head_host.copy_(head, non_blocking=True)
                f.head_ready = torch.cuda.Event()
                f.head_ready.record(self.copy_stream)
        self.slot = (self.slot + 1) % 3
        return f

    def _finish(self, f):
        out = f.out
        with torch.cuda.stream(self.copy_stream):
            if f.head_ready is not None:
                f.head_ready.synchronize()
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
        return out

    def __iter__(self):
        return self

    def __next__(self):
        if self.next is None:
            raise StopIteration
        out = self._finish(self.next)
        self.next = self._start()  # the device works on it beside the scene just handed over
        return out
