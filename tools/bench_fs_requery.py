"""Dev tool: few-shot test loop on one S150k query scene: E sequential cached re-queries vs requery_many."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene
from geoformer_amd.model import GeoFormerFS, load_config
from tests.util import synthetic_state_dict
dev = torch.device("cuda", 0)
mv = lambda d: {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}
cfg = load_config("test_geoformer_fs_scannet.yaml")
m = GeoFormerFS(cfg); m.load_state_dict(synthetic_state_dict(m.state_dict(), 2)); m.semantic_linear.bias.data[3] += 1.0
m.to(dev); m.eval()
def fsd(sc):
    d = scene.make_batch([sc]); d["batch_offsets"] = d["offsets"]; d["support_masks"] = (d["instance_labels"] >= 0).long(); return mv(d)
q = fsd(scene.make_scene(150_000, 1234)); sup = fsd(scene.make_scene(130_000, 70))
E = 40
with torch.no_grad():
    emb = m.process_support(sup, training=False)
    m(None, q, training=False, remember=False, support_embeddings=emb)
    embs = torch.cat([emb * (0.5 + 0.02 * i) for i in range(E)])
    for fn, name in ((lambda: [m(None, q, training=False, remember=True, support_embeddings=embs[i:i+1]) for i in range(E)], "sequential"),
                     (lambda: m.requery_many(q, embs), "requery_many")):
        fn(); torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter()-t)/E*1e3:.3f} ms per re-query ({E} embeddings)")
