"""Serving loop for the eval forward: scenes staggered over two streams on one GPU.

The reference's test loop (test.py:60-110) runs one scene after the other.  On MI355X a scene's forward has two stretches
during which the chip is nearly idle -- ~0.7 ms behind the backbone (the foreground count travels to the host, the host
makes the reference's sampling draw, the first 256 sampling picks run on 16 compute units) and then ~2 ms of sampling
rounds beside the geodesic BFS (one latency-bound workgroup per query) -- and two that need nothing but free compute
units: the backbone's convolutions and the decoder + mask head.  ``StaggeredForward`` fills the former with the latter
ACROSS scenes (device order; lanes are HIP streams, a scene runs on one of them):

    lane A:  bb-B(i) | count, draw, first picks (i) | sampling / BFS (i) | s.a. |                        tail(i) | bb-B(i+2) ..
    lane B:            bb-A(i+1) ..................... | tail(i-1) ......... | bb-B(i+1) | count.. (i+1) | sampling / BFS (i+1) ..

* bb-A = the next scene's voxelisation, rulebooks and first two U-Net levels (gf_unet_fwd_phased), queued by the host
  BEFORE it waits for scene i's foreground count and gated on scene i's backbone event: it runs under scene i's
  read-back, draw and first picks;
* tail = a scene's decoder layers, mask head and proposal statistics, in workgroup shapes that fit on a compute unit
  beside a BFS workgroup (``pointops.co_resident_launches``): it runs under the NEXT scene's sampling / BFS stretch;
* bb-B = the rest of the backbone, held behind the END of the previous scene's stretch (stream events): conv kernels
  beside that stretch slow every one of its sampling rounds and BFS hops by more than they gain.

Host order of ``submit(scene i+1)``: queue bb-A(i+1); [hand-over, called by the native backbone between its phases:
scene i's read-back, draw, sampling / BFS launches and set abstraction; then tail(i-1); then the proposals of scene i-2];
queue bb-B(i+1) and scene i+1's semantic head.  Same operators and values per scene as ``GeoFormer.forward`` (the
cross-attention's 8-wave shape: equal to rounding); every scene handed to ``submit`` is complete when ``drain`` returns.
"""
from __future__ import annotations

import numpy as np
import torch

from . import unet_exec

_LANES = {}


class StaggeredForward:
    def __init__(self, model, device, epoch=300, phased=True):
        self.model, self.device, self.epoch, self.phased = model, torch.device(device), epoch, bool(phased)
        # the two lanes are a process resource (like the forward's side streams, which are keyed by them): a second loop
        # object with fresh streams from the framework's pool has measured 5.0 against 3.9 ms per scene (bench.py's nq = 128
        # leg) -- which hardware queue a stream lands on depends on what was created before it
        key = (self.device.type, self.device.index if self.device.index is not None else torch.cuda.current_device())
        if key not in _LANES:
            _LANES[key] = [torch.cuda.Stream(device=self.device) for _ in range(2)]
        self.lanes = _LANES[key]
        self.n = 0          # scenes submitted so far
        self.head = None    # (SplitForward, lane, seed) of the newest scene: backbone queued, nothing behind it yet
        self.tailq = None   # (SplitForward, lane) of the scene before it: stretch queued, last part not yet
        self.prev = None    # outputs whose proposals are not collected yet
        self._done = []

    def submit(self, batch, seed=None):
        """Queues one scene; returns the outputs of the scenes that completed meanwhile (oldest first, usually one).
        seed: numpy seed set right before THIS scene's sampling draw (which the loop makes one ``submit`` later)."""
        lane = self.lanes[self.n % 2]
        self.n += 1
        self._done = []
        if not self.phased:
            # the plain staggering: this scene's whole backbone behind the end of the previous scene's stretch, then its
            # own read-back / draw / sampling launches, then the previous scene's last part (4.68 against 4.55 ms per scene)
            with torch.cuda.stream(lane), torch.no_grad():
                if self.tailq is not None:
                    for ev in self.tailq[0].stretch_done:
                        lane.wait_event(ev)
                h = self.model.forward_split(batch, self.epoch, training=False, defer_proposals=True)
            self.head = (h, lane, seed)
            self._hand_over()
            return list(self._done)
        gate = [self.head[0].backbone_done] if self.head is not None and self.head[0].backbone_done is not None else []
        with torch.cuda.stream(lane), torch.no_grad():
            unet_exec.phase_next_forward(gate, self._hand_over)
            try:
                h = self.model.forward_split(batch, self.epoch, training=False, defer_proposals=True)
            finally:
                unused = unet_exec.take_phase()
            if unused is not None:  # a backbone that did not go through the native executor: the hand-over comes now
                self._hand_over()
        self.head = (h, lane, seed)
        return list(self._done)

    def _hand_over(self):
        """The newest scene's read-back, draw, sampling / BFS launches and set abstraction; then the last part of the
        scene before it, under that stretch.  Returns the events behind the stretch."""
        if self.head is None:
            return []
        h, lane, seed = self.head
        self.head = None
        if seed is not None:
            np.random.seed(seed)
        with torch.cuda.stream(lane), torch.no_grad():
            h.advance()
        self._tail()
        self.tailq = (h, lane)
        return list(h.stretch_done)

    def _tail(self):
        if self.tailq is not None:
            h, lane = self.tailq
            self.tailq = None
            with torch.cuda.stream(lane), torch.no_grad():
                out = h.finish()
            done = self._collect(self.prev)
            if done is not None:
                self._done.append(done)
            self.prev = out

    @staticmethod
    def _collect(out):
        p = out.get("proposal_scores") if out is not None else None
        if p is not None and not isinstance(p, tuple):
            out["proposal_scores"] = p.get()
        return out

    def drain(self):
        """Queues what is left, collects every pending scene and joins the lanes into the current stream."""
        self._done = []
        self._hand_over()
        self._tail()
        done = self._collect(self.prev)
        self.prev = None
        if done is not None:
            self._done.append(done)
        cur = torch.cuda.current_stream(self.device)
        for lane in self.lanes:
            cur.wait_stream(lane)
        return list(self._done)

    def close(self):
        """Teardown of a serving loop: drain, wait for the device and return the grow-only per-stream scratch blocks
        (pointops.release_scratch: ~0.8 GB per lane after a 150k-point scene) to the allocator.  Returns the scenes the
        drain completed."""
        from . import pointops

        done = list(self.drain())
        torch.cuda.synchronize(self.device)
        pointops.release_scratch(self.device)
        return done
