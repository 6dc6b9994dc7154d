"""Dev tool: where the host is, relative to the device, along one eval forward: host time at the entry of each stage
against the time the device reaches an event recorded there (both from the forward's start)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene
dev = torch.device("cuda", 0)
batches = [bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234 + i)]), dev) for i in range(4)]
model = bench.build_model(dev, probe_batch=batches[0])
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record()
    marks.append((name, time.perf_counter(), e))
def wrap(obj, attr, name):
    f = getattr(obj, attr)
    def w(*a, **k):
        mark(name + " in")
        r = f(*a, **k)
        mark(name + " out")
        return r
    setattr(obj, attr, w)
for attr in ("forward_backbone", "_aggregate_geodesic_overlapped", "forward_decoder", "get_mask_prediction", "generate_proposal"):
    wrap(model, attr, attr)
def run(i):
    np.random.seed(i)
    with torch.no_grad():
        return model(batches[i % 4], 300, training=False, defer_proposals=True)
prev = None
for i in range(8):
    marks.clear()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e0.record(); t0 = time.perf_counter()
    out = run(i)
    mark("returned")
    if prev is not None: prev["proposal_scores"].get()
    prev = out
    torch.cuda.synchronize()
    if i >= 6:
        for name, t, e in marks:
            print(f"{name:42s} host {1e3 * (t - t0):7.3f} ms   device {e0.elapsed_time(e):7.3f} ms")
        print()
