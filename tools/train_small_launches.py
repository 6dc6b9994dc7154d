"""Dev tool: where the training step's SMALL framework launches come from.  One batch-4 step under torch.profiler with
Python stacks; kernels shorter than 12 us that are not this library's are grouped by the innermost frame inside the
repository (forward) or by the autograd node that launched them (backward)."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import geoformer_amd
geoformer_amd.configure_runtime()
import numpy as np, torch
import bench, train_dp
from geoformer_amd import scene, parallel
dev = torch.device("cuda", 0)
mk = lambda seeds: bench.to_device(scene.make_batch([scene.make_scene(int(n), sd) for n, sd in seeds]), dev)
batches = [mk(((150_000, 50), (120_000, 51), (180_000, 52), (100_000, 53)))]
args = train_dp.default_args(steps=1, warmup=2, batch_size=4, epoch=200, prepare_epochs=120, fg_frac=0.4)
cfg, m, crit = train_dp.build(args, dev)
red = parallel.BucketedGradReducer(m, bucket_bytes=int(args.bucket_mb * (1 << 20)))
opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, fused=True)
train_dp.calibrate_foreground(m, batches[0], 0.4)
for i in range(3): train_dp.step(m, crit, red, opt, batches[0], 200, i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_dp.step(m, crit, red, opt, batches[0], 200, 7)
    torch.cuda.synchronize()
ev = prof.events()
kern = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
print("device events", len(kern))
# map kernels to the CPU op that launched them through correlation: use key_averages grouped by stack instead
ka = prof.key_averages(group_by_stack_n=12)
rows = []
for a in ka:
    if a.device_time_total <= 0 or a.count == 0: continue
    rows.append(a)
by = collections.Counter(); bt = collections.Counter()
for a in rows:
    name = a.key
    if not (name.startswith("aten::") ): continue
    frames = [f for f in (a.stack or []) if "/repo/" in f and "tools/" not in f]
    site = frames[0].split("/repo/")[-1] if frames else "(autograd engine / no repo frame)"
    by[(name, site)] += a.count; bt[(name, site)] += a.self_device_time_total
print("aten ops with device time, by (op, innermost repo frame): count, self device us")
for (k, c) in by.most_common(60):
    print(f"{c:5d} {bt[k]:9.0f} us  {k[0]:28s} {k[1][:110]}")
