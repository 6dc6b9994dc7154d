"""Dev tool: full eval forward on the S150k scene with per-stage device timings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene
from geoformer_amd.model import GeoFormer, load_config
from tests.util import synthetic_state_dict

cfg = load_config(sys.argv[1] if len(sys.argv) > 1 else "test_geoformer_scannet.yaml")
m = GeoFormer(cfg)
m.load_state_dict(synthetic_state_dict(m.state_dict(), 0))
m.cuda(); m.eval()
sc = scene.make_scene(150_000, 1234)
batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in scene.make_batch([sc]).items()}
stages = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); stages.setdefault(label, []).append(time.perf_counter() - t)
        return r
    setattr(obj, name, g)
import geoformer_amd.model.geoformer as G
wrap(m, "forward_backbone", "backbone"); wrap(m, "forward_aggregator", "aggregator")
wrap(G, "cal_geodesic", "geodesic"); wrap(m, "forward_decoder", "decoder")
wrap(m, "get_mask_prediction", "mask_head"); wrap(m, "generate_proposal", "proposal")
for it in range(6):
    np.random.seed(it)
    torch.cuda.synchronize(); t = time.perf_counter()
    with torch.no_grad():
        out = m(batch, 300, training=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"iter {it}: {dt*1e3:.1f} ms  N_fg {out['fg_idxs'].shape[0]}  " + "  ".join(f"{k} {v[-1]*1e3:.1f}" for k, v in stages.items()))
print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
