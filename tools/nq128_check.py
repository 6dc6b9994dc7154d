"""Dev tool: the nq=128 (train yaml) eval forward of bench.py's secondary leg alone."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene
dev = torch.device("cuda", 0)
ns = 8
batches = [bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234 + i)]), dev) for i in range(ns)]
model = bench.build_model(dev, probe_batch=batches[0])
for cfg in ("geoformer_scannet.yaml", "test_geoformer_scannet.yaml"):
    m = bench.build_model(dev, bias_shift=model._bench_bias_shift, cfg_name=cfg)
    prev = [None]
    def fin():
        if prev[0] is not None and not isinstance(prev[0].get("proposal_scores"), (tuple, type(None))):
            prev[0]["proposal_scores"] = prev[0]["proposal_scores"].get()
        prev[0] = None
    def step(i):
        np.random.seed(1000 + i)
        with torch.no_grad():
            out = m(batches[i % ns], 300, training=False, defer_proposals=True)
        fin(); prev[0] = out
    for i in range(ns): step(i)
    fin(); torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(16): step(100 + i)
    fin(); torch.cuda.synchronize()
    print(cfg, "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "GF_BFS_WG", os.environ.get("GF_BFS_WG"), "%.3f ms per scene" % ((time.perf_counter() - t) / 16 * 1e3), flush=True)
