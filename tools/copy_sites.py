"""Dev tool: which Python call sites of one eval forward issue tensor copies / fills (candidates for the
__amd_rocclr_copyBuffer / fillBufferAligned launches in the kernel trace)."""
import sys, os, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene
dev = torch.device("cuda", 0)
batch = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
model = bench.build_model(dev, probe_batch=batch)
def step():
    np.random.seed(0)
    with torch.no_grad():
        return model(batch, 300, training=False)
for _ in range(3): step()
torch.cuda.synchronize()
CNT = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "geoformer_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"
def wrap(obj, name, label=None):
    orig = getattr(obj, name)
    def w(*a, **k):
        CNT[(label or name, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, w)
    return orig
T = torch.Tensor
for n in ("copy_", "clone", "to", "contiguous", "item", "tolist", "__setitem__", "zero_", "fill_", "cuda", "cpu", "float", "int", "long"):
    wrap(T, n)
for n in ("tensor", "cat", "stack", "zeros", "full", "zeros_like", "ones_like", "as_tensor", "from_numpy"):
    wrap(torch, n)
step(); torch.cuda.synchronize()
for (op, s), v in sorted(CNT.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{v:4d} {op:14s} {s}")
