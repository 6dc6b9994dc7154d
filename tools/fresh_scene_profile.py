"""Dev tool: what a never-before-seen scene size costs the host: per step wall time, device allocations made by the
caching allocator (hipMalloc calls), and a cProfile of the steps."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene
dev = torch.device("cuda", 0)
b0 = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
model = bench.build_model(dev, probe_batch=b0)
rs = np.random.RandomState(99)
sizes = rs.permutation(np.linspace(0.72, 1.28, 18) * 150_000).astype(int)
fresh = [bench.to_device(scene.make_batch([scene.make_scene(int(n), 5000 + j)]), dev) for j, n in enumerate(sizes)]
def run(b, i):
    np.random.seed(1000 + i)
    torch.cuda.synchronize()
    a0 = torch.cuda.memory_stats()["num_device_alloc"]
    t = time.perf_counter()
    with torch.no_grad():
        model(b, 300, training=False)
    th = time.perf_counter() - t
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3, th * 1e3, torch.cuda.memory_stats()["num_device_alloc"] - a0
for i in range(4): run(b0, i)
print("repeat scene:", ["%.2f/%.2f ms, %d mallocs" % run(b0, i) for i in range(3)])
pr = cProfile.Profile()
for j, b in enumerate(fresh):
    if j >= 6: pr.enable()
    r = run(b, j)
    if j >= 6: pr.disable()
    print("fresh %2d n=%6d: total %.2f ms, host returns after %.2f ms, %d device allocations; again: %.2f ms" % ((j, sizes[j]) + r + (run(b, j)[0],)))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
