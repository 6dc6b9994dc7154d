"""Dev tool: cProfile of the host side of one eval forward (S150k)."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import geoformer_amd
geoformer_amd.configure_runtime()
import numpy as np, torch
import bench
from geoformer_amd import scene
dev = torch.device("cuda", 0)
batch = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
model = bench.build_model(dev, probe_batch=batch)
def step():
    np.random.seed(0)
    with torch.no_grad():
        return model(batch, 300, training=False)
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(60); print(s.getvalue()[:9000])

import time
th = tt = 0.0
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    th += t1 - t0; tt += t2 - t0
print(f"host returns after {th / 20 * 1e3:.2f} ms, device done after {tt / 20 * 1e3:.2f} ms")
