"""Serving loop for the eval forward: two scenes in flight on one GPU, staggered.

The reference's test loop (test.py:60-110) runs one scene after the other.  On MI355X a scene's forward has a stretch of
~2 ms -- 2047 serial sampling rounds on 16 compute units beside the geodesic BFS, one latency-bound workgroup per query --
during which the chip is nearly idle, and the decoder + mask head that follow are matrix work that needs nothing but
free compute units.  ``StaggeredForward`` overlaps the two ACROSS scenes:

    lane A:  backbone(i) | sampling / BFS (i) | set abstr. |                  decoder + mask head (i) | backbone(i+2) ...
    lane B:                                     backbone(i+1) | sampling / BFS (i+1) .................. | set abstr. | ...
                                                ^ starts when stretch(i) has ended

* a scene's backbone waits for the END of the previous scene's sampling / BFS stretch (stream events): conv kernels
  beside that stretch slow every one of its sampling rounds and BFS hops by more than they gain;
* scene i's decoder / mask head / proposal statistics are queued by the host right after scene i+1's sampling and BFS
  launches and execute under that stretch, in workgroup shapes that fit on a compute unit beside a BFS workgroup
  (``pointops.co_resident_launches``: the 16-wave cross-attention does not, it waited for the BFS queries to retire);
* the host collects scene i-1's proposals (one pinned word, long there) after that.

Same operators and values as ``GeoFormer.forward`` (the cross-attention's 8-wave shape: equal to rounding); every scene
handed to ``submit`` is complete when ``drain`` returns.  Measured and dropped: holding scene i's decoder behind scene
i+1's BACKBONE with a device-side gate so that it could be queued earlier (a polling one-wave kernel: every launch of
the other stream then started ~60 us after the previous one; hipStreamWaitValue32: the streams stopped for good --
the gated lane shares a hardware queue with a stream the other lane's backbone needs).
"""
from __future__ import annotations

import collections

import torch



class StaggeredForward:
    def __init__(self, model, device, epoch=300):
        self.model, self.device, self.epoch = model, torch.device(device), epoch
        self.lanes = [torch.cuda.Stream(device=self.device) for _ in range(2)]
        self.n = 0          # scenes submitted so far
        self.stretch = ()   # events behind the newest scene's sampling / BFS stretch
        self.head = None    # (SplitForward, lane) of the newest scene: its last part is not queued yet
        self.prev = None    # outputs whose proposals are not collected yet

    def submit(self, batch):
        """Queues one scene; returns the outputs of the scene that completed meanwhile (or None)."""
        lane = self.lanes[self.n % 2]
        self.n += 1
        with torch.cuda.stream(lane), torch.no_grad():
            for ev in self.stretch:
                lane.wait_event(ev)
            h = self.model.forward_split(batch, self.epoch, training=False, defer_proposals=True)
            h.advance()
        self.stretch = h.stretch_done
        done = self._tail()
        self.head = (h, lane)
        return done

    def _tail(self):
        """The newest scene's last part (under the stretch just queued), then the proposals of the scene before it."""
        done = None
        if self.head is not None:
            h, lane = self.head
            self.head = None
            with torch.cuda.stream(lane), torch.no_grad():
                out = h.finish()
            done = self._collect(self.prev)
            self.prev = out
        return done

    @staticmethod
    def _collect(out):
        p = out.get("proposal_scores") if out is not None else None
        if p is not None and not isinstance(p, tuple):
            out["proposal_scores"] = p.get()
        return out

    def drain(self):
        """Queues what is left, collects every pending scene and joins the lanes into the current stream."""
        done = [o for o in (self._tail(), self._collect(self.prev)) if o is not None]
        self.prev = None
        self.stretch = ()
        cur = torch.cuda.current_stream(self.device)
        for lane in self.lanes:
            cur.wait_stream(lane)
        return done
