"""Dev tool: where do the GPU forward and the oracle-backed host forward of the S150k scene drift apart?
Max-abs difference and value scale after every U-Net stage (hooks on blocks / blocks_tail / UBlock outputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_model, to_device
from geoformer_amd import scene
from oracle import cpu_backend

pts = int(sys.argv[1]) if len(sys.argv) > 1 else 150_000
batch = scene.make_batch([scene.make_scene(pts, 1234)])
dev_batch = to_device(batch, "cuda")
m = build_model("cuda", probe_batch=dev_batch)

def hook_all(model, store):
    hs = []
    def add(mod, name):
        def h(_m, _i, o):
            f = o.features if hasattr(o, "features") else o
            store[name] = f.detach().double().cpu().numpy().copy()
        hs.append(mod.register_forward_hook(h))
    add(model.input_conv, "input_conv")
    u, lvl = model.unet, 1
    while u is not None:
        add(u.blocks, f"L{lvl}.blocks")
        if hasattr(u, "blocks_tail"):
            add(u.blocks_tail, f"L{lvl}.tail")
        add(u, f"L{lvl}.out")
        u, lvl = getattr(u, "u", None), lvl + 1
    return hs

g, c = {}, {}
hs = hook_all(m, g)
with torch.no_grad():
    og = m(dev_batch, 0, training=False)
torch.cuda.synchronize()
for h in hs: h.remove()
with cpu_backend.installed(), torch.no_grad():
    mc = build_model("cpu", bias_shift=m._bench_bias_shift)
    hook_all(mc, c)
    oc = mc(batch, 0, training=False)
for k in c:
    if k in g:
        d = np.abs(g[k] - c[k])
        print(f"{k:14s} shape {str(g[k].shape):16s} maxabs {d.max():.3e} mean {d.mean():.3e} scale {np.abs(c[k]).max():.3e} rms {np.sqrt((c[k]**2).mean()):.3e}")
d = (og["semantic_scores"].cpu() - oc["semantic_scores"]).abs()
print("semantic maxabs", float(d.max()), "scale", float(oc["semantic_scores"].abs().max()))
