"""Dev tool: host issue time against GPU completion time of the backbone stretch and of the whole eval forward
(S150k, rotating scenes): is a stretch bound by the host's launch rate or by the device?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene
dev = torch.device("cuda", 0)
batches = [bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234 + i)]), dev) for i in range(4)]
model = bench.build_model(dev, probe_batch=batches[0])
def backbone(b):
    with torch.no_grad():
        return model.forward_backbone(b, 1, want_preds=False)
def full(b):
    np.random.seed(0)
    with torch.no_grad():
        return model(b, 300, training=False)
for fn, name in ((backbone, "backbone"), (full, "forward")):
    for i in range(6): fn(batches[i % 4])
    torch.cuda.synchronize()
    hi, tot = [], []
    for i in range(16):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(batches[i % 4]); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        hi.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
    print(f"{name}: host issue {np.mean(hi):.3f} ms  until device done {np.mean(tot):.3f} ms")
    # back to back (no sync between steps)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(16): fn(batches[i % 4])
    torch.cuda.synchronize(); print(f"{name}: back to back {(time.perf_counter() - t0) / 16 * 1e3:.3f} ms per call")
