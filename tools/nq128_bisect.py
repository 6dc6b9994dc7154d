"""Dev tool: which of bench.py's stages before the nq=128 leg makes that leg slow (16 hardware queues)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene
dev = torch.device("cuda", 0)
ns = 8
batches = [bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234 + i)]), dev) for i in range(ns)]
model = bench.build_model(dev, probe_batch=batches[0])
prev = [None]
def fin():
    if prev[0] is not None and not isinstance(prev[0].get("proposal_scores"), (tuple, type(None))):
        prev[0]["proposal_scores"] = prev[0]["proposal_scores"].get()
    prev[0] = None
def run(m, tag):
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
    print(tag, "%.3f ms per scene" % ((time.perf_counter() - t) / 16 * 1e3), flush=True)
def m128():
    return bench.build_model(dev, bias_shift=model._bench_bias_shift, cfg_name="geoformer_scannet.yaml")
run(model, "main model")
m = m128(); run(m, "nq128 after main loop"); del m
probe = bench.ConvProbe(batches)
for i in range(8):
    probe.arm(i % 4 == 0)
    np.random.seed(i)
    with torch.no_grad():
        model(batches[i % ns], 300, training=False)
probe.close(); probe.result()
m = m128(); run(m, "nq128 after ConvProbe"); del m
bench.all_convs_roofline(model, batches)
m = m128(); run(m, "nq128 after all_convs_roofline"); del m
bench.op_rooflines(model, batches)
m = m128(); run(m, "nq128 after op_rooflines"); run(m, "nq128 again"); del m
run(model, "main model again")
