"""Dev tool: run the backbone stretch a few times (for a rocprofv3 kernel trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene
dev = torch.device("cuda", 0)
batches = [bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234 + i)]), dev) for i in range(4)]
model = bench.build_model(dev, probe_batch=batches[0])
for i in range(12):
    with torch.no_grad():
        model.forward_backbone(batches[i % 4], 1, want_preds=False)
    torch.cuda.synchronize()
