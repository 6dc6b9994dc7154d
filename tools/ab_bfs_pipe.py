"""Dev tool: A/B of the two BFS kernels (gf_dev_bfs_pipe 0 / 1) inside ONE process: eval forwards over the eight benchmark
scenes alternate between the kernels, each timed from launch to completion; per-scene paired differences."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene, _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
batches = [bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234 + i)]), dev) for i in range(8)]
model = bench.build_model(dev, probe_batch=batches[0])
def step(i, ab):
    lib.gf_dev_bfs_pipe(ab)
    np.random.seed(1000 + i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.no_grad():
        model(batches[i % 8], 300, training=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3
for i in range(16): step(i, i % 2)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
t = {0: [], 1: []}
for i in range(n):
    for ab in ((0, 1) if (i // 8) % 2 == 0 else (1, 0)):
        t[ab].append(step(i, ab))
a, b = np.array(t[0]), np.array(t[1])
print("lds median %.3f ms   pipe median %.3f ms   paired diff (pipe - lds) median %+.3f ms  mean %+.3f" % (np.median(a), np.median(b), np.median(b - a), np.mean(b - a)))
for s in range(8):
    print("  scene %d: lds %.3f  pipe %.3f  diff %+.3f" % (s, np.median(a[s::8]), np.median(b[s::8]), np.median((b - a)[s::8])))
lib.gf_dev_bfs_pipe(-1)
