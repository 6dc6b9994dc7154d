"""Dev tool: is bench.py's fresh_scenes leg slower than the headline because the scenes are FRESH (first time a size is
seen) or because of the scene mix?  The same 24 scenes three times over; the first pass is the benchmark's leg."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import geoformer_amd
geoformer_amd.configure_runtime()
import numpy as np, torch
import bench
from geoformer_amd import scene
dev = torch.device("cuda", 0)
P = 150_000
probe = bench.to_device(scene.make_batch([scene.make_scene(P, 1234)]), dev)
model = bench.build_model(dev, probe_batch=probe)
heads = [bench.to_device(scene.make_batch([scene.make_scene(P, 1234 + j)]), dev) for j in range(8)]
rs = np.random.RandomState(99)
sizes = rs.permutation(np.linspace(0.72, 1.28, 26) * P).astype(int)
fresh = [bench.to_device(scene.make_batch([scene.make_scene(int(n), 5000 + j)]), dev) for j, n in enumerate(sizes)]
def run(scenes, idx):
    prev = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in idx:
        np.random.seed(1000 + i)
        with torch.no_grad():
            out = model(scenes[i % len(scenes)], 300, training=False, defer_proposals=True)
        if prev is not None: prev["proposal_scores"] = prev["proposal_scores"].get()
        prev = out
    prev["proposal_scores"] = prev["proposal_scores"].get()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / len(idx) * 1e3
run(heads, range(8)); run(heads, range(8))
print(f"headline scenes: {run(heads, range(24)):.3f} ms per scene")
model.reserve_for(int(os.environ.get("RESERVE", int(1.3 * P))))
run(fresh, [0, 1])
for k in range(3):
    st0 = torch.cuda.memory_stats()
    t = run(fresh, range(2, 26))
    st1 = torch.cuda.memory_stats()
    print(f"fresh set, pass {k + 1}: {t:.3f} ms per scene; device mallocs {st1['num_device_alloc'] - st0['num_device_alloc']}, "
          f"frees {st1['num_device_free'] - st0['num_device_free']}, segments {st1['segment.all.allocated'] - st0['segment.all.allocated']}, "
          f"reserved {st1['reserved_bytes.all.current'] / 2**30:.2f} GiB")
per = []
for i in range(2, 26):
    per.append((int(fresh[i]["locs"].shape[0]), run(fresh, [i, i, i])))
print("per scene (points, ms):", " ".join(f"{n}:{t:.2f}" for n, t in sorted(per)))
print(f"mean of per-scene times {np.mean([t for _, t in per]):.3f} ms")
