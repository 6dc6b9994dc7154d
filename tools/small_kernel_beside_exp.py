"""Dev tool (experiment): what does a chain of small kernels cost beside the sampling kernel, beside the BFS, beside both?
(In the staggered loop the decoder's token stages take ~55 us each under the next scene's stretch, 18-23 us alone.)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene, pointops
dev = torch.device("cuda", 0)
batch = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
model = bench.build_model(dev, probe_batch=batch)
NAMES = ("geodesic_bfs", "furthest_point_sampling")
calls = []
saved = {n: getattr(pointops, n) for n in NAMES}
def wrap(n, fn):
    def w(*a, **k):
        calls.append((n, a, k)); return fn(*a, **k)
    return w
for n in NAMES: setattr(pointops, n, wrap(n, saved[n]))
np.random.seed(1002)
with torch.no_grad(): model(batch, 300, training=False)
torch.cuda.synchronize()
for n in NAMES: setattr(pointops, n, saved[n])
fps = [c for c in calls if c[0] == "furthest_point_sampling"][-1:]
bfs = [c for c in calls if c[0] == "geodesic_bfs"]
s = [torch.cuda.Stream() for _ in range(3)]
x = torch.randn(256, 64, device=dev); y = torch.empty_like(x)
NK = 24
def chain():
    for _ in range(NK): torch.add(x, 1.0, out=y)
def play(cs, st):
    with torch.cuda.stream(st):
        for n, a, k in cs: saved[n](*a, **k)
def run(do_fps, do_bfs, do_chain):
    main = torch.cuda.current_stream()
    for q in s: q.wait_stream(main)
    if do_fps: play(fps, s[0])
    if do_bfs: play(bfs, s[1])
    e0 = e1 = None
    if do_chain:
        with torch.cuda.stream(s[2]):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); chain(); e1.record()
    for q in s: main.wait_stream(q)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / NK if do_chain else None
for label, f, b in (("alone", 0, 0), ("beside the sampling", 1, 0), ("beside the BFS", 0, 1), ("beside both", 1, 1)):
    run(f, b, 1)
    v = [run(f, b, 1) for _ in range(5)]
    print("%-22s %.1f us per small kernel (start to start)" % (label, float(np.median(v))))
