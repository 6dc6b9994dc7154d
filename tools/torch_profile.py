"""Dev tool: torch.profiler over one eval forward -- which aten ops issue memcpy / memset (by call stack)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
batch = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
model = bench.build_model(dev, probe_batch=batch)
def step():
    np.random.seed(0)
    with torch.no_grad():
        return model(batch, 300, training=False)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
ev = prof.events()
import collections
cnt = collections.Counter()
for e in ev:
    if e.device_type.name == "CPU" and e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::item", "aten::_local_scalar_dense", "aten::nonzero"):
        st = [s for s in (e.stack or []) if "geoformer_amd" in s or "bench.py" in s]
        cnt[(e.name, st[0].split("/")[-1][:70] if st else "?")] += 1
for k, v in cnt.most_common(40): print(v, k)
