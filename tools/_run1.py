import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from geoformer_amd import scene, pointops
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): out = fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
p = scene.make_scene(150_000, 1234)["xyz"]
for n in (50000, 40000):
    idx = np.random.default_rng(1).permutation(p.shape[0])[:n]
    xyz = torch.from_numpy(np.ascontiguousarray(p[idx])).cuda()[None].contiguous()
    print(n, "2048 picks: %.1f us" % timeit(lambda: pointops.furthest_point_sampling(xyz, 2048)), " 256 picks: %.1f us" % timeit(lambda: pointops.furthest_point_sampling(xyz, 256)))
