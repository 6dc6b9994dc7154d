"""Dev tool: eval forwards over the eight benchmark scenes with whatever library GF_LIB_PATH names: per-scene medians
(for A/B of build variants across two runs on ONE box: run both in the same gpurun call)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import geoformer_amd
geoformer_amd.configure_runtime()
import numpy as np, torch
import bench
from geoformer_amd import scene
dev = torch.device("cuda", 0)
batches = [bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234 + i)]), dev) for i in range(8)]
model = bench.build_model(dev, probe_batch=batches[0])
def step(i):
    np.random.seed(1000 + i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.no_grad():
        model(batches[i % 8], 300, training=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3
for i in range(16): step(i)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
t = np.array([step(i) for i in range(n)])
print(os.environ.get("GF_LIB_PATH", "default"), "median %.3f ms  mean %.3f  per scene: %s" % (np.median(t), t.mean(), " ".join("%.2f" % np.median(t[s::8]) for s in range(8))))
