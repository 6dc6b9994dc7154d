"""Dev tool: cProfile of the host side of the staggered serving loop (which Python frames cost the ~2.7 ms per scene)."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene, serving
dev = torch.device("cuda", 0)
ns = 4
batches = [bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234 + i)]), dev) for i in range(ns)]
model = bench.build_model(dev, probe_batch=batches[0])
loop = serving.StaggeredForward(model, dev)
for i in range(8): loop.submit(batches[i % ns], seed=i)
loop.drain(); torch.cuda.synchronize()
N = 40
t = time.perf_counter()
for i in range(N): loop.submit(batches[i % ns], seed=i)
loop.drain(); torch.cuda.synchronize()
print("%.3f ms per scene" % ((time.perf_counter() - t) / N * 1e3))
pr = cProfile.Profile(); pr.enable()
for i in range(N): loop.submit(batches[i % ns], seed=i)
loop.drain(); torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumulative"):
    sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats(key).print_stats(38)
    print("\n".join(l[:150] for l in sio.getvalue().splitlines()[:60]))
