"""Dev tool: A/B of a code path inside ONE process: forwards alternate between GF_AB=0 and GF_AB=1 (any code may read
that variable per call), each timed from launch to completion; medians and the paired difference are printed.  Removes
the box-to-box and run-to-run spread of bench.py (about +-80 us) from the comparison."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene
dev = torch.device("cuda", 0)
batch = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
model = bench.build_model(dev, probe_batch=batch)
def step(i, ab):
    os.environ["GF_AB"] = ab
    np.random.seed(1000 + i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.no_grad():
        model(batch, 300, training=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3
for i in range(6): step(i, "01"[i % 2])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
t = {"0": [], "1": []}
for i in range(n):
    for ab in (("0", "1") if i % 2 == 0 else ("1", "0")):
        t[ab].append(step(i, ab))
a, b = np.array(t["0"]), np.array(t["1"])
print("GF_AB=0 median %.3f ms   GF_AB=1 median %.3f ms   paired diff (1-0) median %+.3f ms  mean %+.3f" % (np.median(a), np.median(b), np.median(b - a), np.mean(b - a)))
