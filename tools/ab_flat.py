"""Dev tool: eval forwards alternating between the flat-chain deep-level conv kernel off (A) and size-based (B) in one
process (paired difference; see tools/ab_inprocess.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene, sparse
dev = torch.device("cuda", 0)
batch = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
model = bench.build_model(dev, probe_batch=batch)
KN = {"0": dict(flat=0), "1": {}}
if len(sys.argv) > 2:
    KN["1"] = eval(sys.argv[2])
def step(i, ab):
    sparse.dev_conv_knobs(**KN[ab])
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
print("A %s median %.3f ms   B %s median %.3f ms   paired diff (B-A) median %+.3f ms  mean %+.3f" % (KN["0"], np.median(a), KN["1"], np.median(b), np.median(b - a), np.mean(b - a)))
sparse.dev_conv_knobs()
