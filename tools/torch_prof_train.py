"""Dev tool: torch.profiler over the config-3 training step: framework kernels attributed to Python source lines."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import ProfilerActivity, profile
from geoformer_amd import scene
from geoformer_amd.model import GeoFormer, InstSetCriterion, load_config
from tests.util import synthetic_state_dict

dev = torch.device("cuda", 0)
mv = lambda d: {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}
cfg = load_config("geoformer_scannet.yaml", batch_size=4, prepare_epochs=120)
m = GeoFormer(cfg); m.load_state_dict(synthetic_state_dict(m.state_dict(), 0)); m.to(dev); m.train()
crit = InstSetCriterion(cfg)
opt = torch.optim.Adam(filter(lambda p: p.requires_grad, m.parameters()), lr=1e-3, fused=True)
batch = mv(scene.make_batch([scene.make_scene(int(n), 50 + i) for i, n in enumerate((150_000, 120_000, 180_000, 100_000))]))
def step():
    np.random.seed(0)
    out = m(batch, 200)
    loss, _ = crit(out, batch, 200)
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
def dev_t(e):
    return getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
for op in ("aten::copy_", "aten::sum", "aten::mm", "aten::add", "aten::add_", "aten::fill_", "aten::index", "aten::_index_put_impl_",
           "aten::mul", "aten::bmm", "aten::div", "aten::addmm", "aten::cat", "aten::zero_", "aten::zeros_like"):
    rows = sorted([e for e in ka if e.key == op], key=lambda e: -dev_t(e))
    tot = sum(dev_t(e) for e in rows)
    print(f"== {op}: {tot/1e3:.3f} ms in {sum(e.count for e in rows)} calls")
    for e in rows[:7]:
        print(f"   {dev_t(e)/1e3:7.3f} ms n={e.count:3d} {str(e.input_shapes)[:150]}")
