"""Dev tool: in-radius degree of the kNN rows of the forward's foreground graphs, by scene size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import geoformer_amd
geoformer_amd.configure_runtime()
import numpy as np, torch
import bench
from geoformer_amd import scene, pointops
dev = torch.device("cuda", 0)
probe = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
m = bench.build_model(dev, probe_batch=probe)
cap = []
orig = pointops.geodesic_bfs
def spy(D, I, deg, src, radius, max_step, wg_threads=1024):
    cap.append((D, I, deg)); return orig(D, I, deg, src, radius, max_step, wg_threads=wg_threads)
pointops.geodesic_bfs = spy
for n, sd in ((150_000, 1234), (174_852, 5001), (192_169, 5015), (108_214, 5010)):
    b = bench.to_device(scene.make_batch([scene.make_scene(n, sd)]), dev)
    cap.clear(); np.random.seed(7)
    with torch.no_grad(): out = m(b, 300, training=False)
    torch.cuda.synchronize()
    D, I, deg = cap[0]
    val = ((D <= 0.05) & (I >= 0)).sum(1).float() - 1  # without the self entry
    ext = b["locs_float"].max(0)[0] - b["locs_float"].min(0)[0]
    print(f"points {n}: foreground {D.shape[0]}, in-radius neighbours per row mean {val.mean().item():.1f}, median {val.median().item():.0f}, "
          f"rows with more than 15: {(val > 15).float().mean().item():.2f}, more than 31: {(val > 31).float().mean().item():.2f}; extent {[round(float(x), 1) for x in ext]}")
