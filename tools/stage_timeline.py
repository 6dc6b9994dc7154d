"""Dev tool: per-stage host-issue time vs GPU time of the eval forward (S150k), no syncs added.
Host time = perf_counter between entry and exit of the stage; GPU time = event pair recorded at the same
places.  A stage whose host time exceeds its GPU time is launch-bound."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene
from geoformer_amd.model import geoformer as G

dev = torch.device("cuda", 0)
batch = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
model = bench.build_model(dev, probe_batch=batch)
REC = []

def wrap(obj, name, label=None):
    label = label or name
    fn = getattr(obj, name)
    if isinstance(fn, torch.nn.Module):
        obj, name, fn = fn, "forward", fn.forward
    def w(*a, **k):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); t0 = time.perf_counter()
        r = fn(*a, **k)
        t1 = time.perf_counter(); e1.record()
        REC.append((label, t0, t1, e0, e1))
        return r
    setattr(obj, name, w)

for n in ["preprocess_input", "prebuild_rulebooks", "input_conv", "unet", "output_layer", "semantic", "mask_tower", "forward_aggregator",
          "forward_decoder", "get_mask_prediction", "generate_proposal", "relative_position_embedding"]:
    wrap(model, n)
wrap(model.set_aggregator, "group_points", "  sa.group_points")
wrap(model.set_aggregator, "mlp", "  sa.mlp")
wrap(model, "decoder", "  decoder")
wrap(G, "cal_geodesic")

def step():
    np.random.seed(0)
    with torch.no_grad():
        return model(batch, 300, training=False)
for _ in range(4): step()
torch.cuda.synchronize()
REC.clear()
N = 10
T0 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize()
T1 = time.perf_counter()
agg = {}
order = []
for lab, t0, t1, e0, e1 in REC:
    if lab not in agg: agg[lab] = [0.0, 0.0]; order.append(lab)
    agg[lab][0] += (t1 - t0) * 1e3; agg[lab][1] += e0.elapsed_time(e1)
print(f"step {1e3*(T1-T0)/N:.2f} ms")
print(f"{'stage':34s} host ms   gpu-span ms")
for lab in order:
    print(f"{lab:34s} {agg[lab][0]/N:7.2f}  {agg[lab][1]/N:7.2f}")
