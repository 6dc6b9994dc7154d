"""Dev tool: N eval forwards of the benchmark model on benchmark scene 0 (S150k, test yaml) and nothing else -- the
workload of tools/pmc_forward.sh (rocprofv3 --pmc passes) and of kernel traces of single forwards."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from geoformer_amd import scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
batch = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
m = bench.build_model(dev, probe_batch=batch)
for i in range(n):
    np.random.seed(1000 + i)
    with torch.no_grad():
        m(batch, 300, training=False)
torch.cuda.synchronize()
print("done")
