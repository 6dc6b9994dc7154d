import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene
from geoformer_amd.model import GeoFormer, InstSetCriterion, load_config
from tests.util import synthetic_state_dict
dev = torch.device("cuda", 0)
mv = lambda d: {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}
cfg = load_config("geoformer_scannet.yaml", batch_size=4, prepare_epochs=120)
m = GeoFormer(cfg); m.load_state_dict(synthetic_state_dict(m.state_dict(), 0)); m.to(dev); m.train()
crit = InstSetCriterion(cfg)
opt = torch.optim.Adam(filter(lambda p: p.requires_grad, m.parameters()), lr=1e-3)
batch = mv(scene.make_batch([scene.make_scene(int(n), 50 + i) for i, n in enumerate((150_000, 120_000, 180_000, 100_000))]))
T = {}
def tic(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); T[name] = T.get(name, 0) + time.perf_counter() - t; return r
def step():
    np.random.seed(0)
    out = tic("forward", lambda: m(batch, 200))
    loss, _ = tic("criterion", lambda: crit(out, batch, 200))
    opt.zero_grad(); tic("backward", lambda: loss.backward()); tic("adam", lambda: opt.step())
step(); T.clear()
import geoformer_amd.model.geoformer as G
for name in ("forward_backbone", "forward_aggregator", "forward_decoder", "get_mask_prediction"):
    f = getattr(m, name)
    setattr(m, name, (lambda f, name: lambda *a, **k: tic("  fwd." + name, lambda: f(*a, **k)))(f, name))
g = G.cal_geodesic; G.cal_geodesic = lambda *a, **k: tic("  fwd.geodesic", lambda: g(*a, **k))
for _ in range(2): step()
for k, v in T.items(): print(f"{k:28s} {v/2*1e3:8.1f} ms")
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
np.random.seed(0); out = m(batch, 200); torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
