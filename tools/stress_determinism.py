"""Dev tool: the same seeded forward many times; every output must equal the first run bit for bit (races between the
three streams of the forward would show up as differences)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene
dev = torch.device("cuda", 0)
batch = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
model = bench.build_model(dev, probe_batch=batch)
def run():
    np.random.seed(7)
    with torch.no_grad():
        out = model(batch, 300, training=False)
    mp = out["mask_predictions"][-1]
    cls, sc, pr = out["proposal_scores"]
    return [out["semantic_scores"], mp["cls_logits"], mp["mask_logits"][0], sc, pr.sum(1)]
ref = [t.clone() for t in run()]
bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for i in range(n):
    got = run()
    for k, (a, b) in enumerate(zip(ref, got)):
        if a.shape != b.shape or not torch.equal(a, b):
            bad += 1
            print("iteration", i, "output", k, "differs: max abs", float((a.float() - b.float()).abs().max()) if a.shape == b.shape else "shape")
            break
print("forwards", n, "mismatching", bad, "nan in logits", bool(torch.isnan(ref[2]).any()))
# fresh model instances: the FIRST forward derives every cached copy (folded BN, packed weights, MLP chains) and must
# already be right
def outputs(model):
    np.random.seed(7)
    with torch.no_grad():
        out = model(batch, 300, training=False)
    mp = out["mask_predictions"][-1]
    return {"semantic_scores": out["semantic_scores"], "fg_idxs": out["fg_idxs"], "cls_logits": mp["cls_logits"],
            "mask_logits": mp["mask_logits"][0]}
fresh_bad = 0
for i in range(int(os.environ.get("FRESH", "12"))):
    model = bench.build_model(dev, probe_batch=None)
    a = {k: v.clone() for k, v in outputs(model).items()}
    b = outputs(model)
    diff = [k for k in a if a[k].shape != b[k].shape or not torch.equal(a[k], b[k])]
    if diff:
        fresh_bad += 1
        print("fresh model", i, ": first and second forward differ in", diff,
              [float((a[k].float() - b[k].float()).abs().max()) for k in diff if a[k].shape == b[k].shape])
print("fresh models with a wrong first forward:", fresh_bad)
