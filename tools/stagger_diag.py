"""Dev tool: host timestamps inside the staggered serving loop (geoformer_amd/serving.py) for a few scenes: when does
the host enter the hand-over, when does the foreground count arrive, when are the sampling launches queued -- against the
device-side end of the scene's backbone (a timing event recorded next to SplitForward.backbone_done)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene, serving, pointops
from geoformer_amd.model import geoformer as gfm
dev = torch.device("cuda", 0)
ns = 4
batches = [bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234 + i)]), dev) for i in range(ns)]
model = bench.build_model(dev, probe_batch=batches[0])
T = time.perf_counter
log = []
cur = {}
orig_get = pointops.PendingForeground.get
def get(self):
    cur["t_get0"] = T(); r = orig_get(self); cur["t_get1"] = T(); return r
pointops.PendingForeground.get = get
orig_fps = pointops.furthest_point_sampling
def fps(*a, **k):
    if "t_fps" not in cur: cur["t_fps"] = T()
    return orig_fps(*a, **k)
pointops.furthest_point_sampling = fps
orig_ho = serving.StaggeredForward._hand_over
def ho(self):
    global cur
    cur = {"t_ho0": T(), "scene": self.n - 2}
    if self.head is not None:
        h = self.head[0]
        cur["bb_ev"] = getattr(h, "_bb_t", None)
    r = orig_ho(self)
    cur["t_ho1"] = T()
    log.append(cur)
    return r
serving.StaggeredForward._hand_over = ho
# a timing event next to backbone_done
orig_init = gfm.SplitForward.__init__
def init(self, steps):
    orig_init(self, steps)
    e = torch.cuda.Event(enable_timing=True); e.record(); self._bb_t = e
gfm.SplitForward.__init__ = init
loop = serving.StaggeredForward(model, dev)
for i in range(6): loop.submit(batches[i % ns], seed=1000 + i)
loop.drain(); torch.cuda.synchronize(); log.clear()
base_e = torch.cuda.Event(enable_timing=True); base_e.record(); torch.cuda.synchronize(); base_h = T()
t_sub = []
for i in range(10):
    t_sub.append(T()); loop.submit(batches[i % ns], seed=2000 + i)
loop.drain(); torch.cuda.synchronize()
for r in log[3:9]:
    h = lambda k: (r[k] - base_h) * 1e3 if k in r else float("nan")
    bb = base_e.elapsed_time(r["bb_ev"]) if r.get("bb_ev") is not None else float("nan")
    print("scene %2d: device backbone+fg-select end %.2f | host: hand-over entered %.2f  count wait %.2f..%.2f  first sampling launch %.2f  hand-over left %.2f" % (
        r["scene"], bb, h("t_ho0"), h("t_get0"), h("t_get1"), h("t_fps"), h("t_ho1")))
print("submit starts:", " ".join("%.2f" % ((t - base_h) * 1e3) for t in t_sub))
