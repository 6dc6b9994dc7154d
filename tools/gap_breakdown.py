"""Dev tool: host-side timeline of the stretch between the backbone and the first sampling launch of an eval forward
(S150k benchmark scene): when the foreground count is back, what the host draw takes, when the sampling launch is issued.
Medians over N forwards, microseconds relative to the count's arrival."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import pointops, scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda", 0)
batch = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
m = bench.build_model(dev, probe_batch=batch)
marks = {}
now = time.perf_counter

def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        marks.setdefault(tag + "_in", now())
        r = f(*a, **k)
        marks.setdefault(tag + "_out", now())
        return r
    setattr(obj, name, g)

wrap(pointops.PendingForeground, "wait", "count")
wrap(pointops, "legacy_prefetch", "prefetch")
wrap(pointops, "draw_sample", "draw")
wrap(pointops, "furthest_point_sampling", "fps")
wrap(pointops, "geodesic_bfs", "bfs")
rows = []
for i in range(n + 4):
    marks.clear()
    np.random.seed(1000 + i)
    torch.cuda.synchronize()
    t0 = now()
    with torch.no_grad():
        m(batch, 300, training=False)
    torch.cuda.synchronize()
    t1 = now()
    if i >= 4:
        z = marks["count_out"]
        rows.append({k: (v - z) * 1e6 for k, v in marks.items()} | {"forward_ms": (t1 - t0) * 1e3, "start": (t0 - z) * 1e6})
keys = ["start", "prefetch_in", "prefetch_out", "count_in", "count_out", "draw_in", "draw_out", "fps_in", "fps_out", "bfs_in", "bfs_out", "forward_ms"]
for k in keys:
    v = [r[k] for r in rows if k in r]
    if v:
        print(f"{k:14s} median {np.median(v):9.1f}   min {min(v):9.1f}   max {max(v):9.1f}")
