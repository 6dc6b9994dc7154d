"""Dev tool: the calibrated batch-4 training step of bench.py's secondary.train_step_b4 (40 % foreground), a few steps,
for rocprofv3 --kernel-trace."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import geoformer_amd
geoformer_amd.configure_runtime()
import numpy as np, torch
import bench, train_dp
from geoformer_amd import scene, parallel
dev = torch.device("cuda", 0)
mk = lambda seeds: bench.to_device(scene.make_batch([scene.make_scene(int(n), sd) for n, sd in seeds]), dev)
batch = mk(((150_000, 50), (120_000, 51), (180_000, 52), (100_000, 53)))
args = train_dp.default_args(steps=1, warmup=2, batch_size=4, epoch=200, prepare_epochs=120, fg_frac=0.4)
cfg, m, crit = train_dp.build(args, dev)
red = parallel.BucketedGradReducer(m, bucket_bytes=int(args.bucket_mb * (1 << 20)))
opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, fused=True)
train_dp.calibrate_foreground(m, batch, 0.4)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for i in range(3): train_dp.step(m, crit, red, opt, batch, 200, i)
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(n): train_dp.step(m, crit, red, opt, batch, 200, 10 + i)
torch.cuda.synchronize(); print(f"step {(time.perf_counter() - t) / n * 1e3:.1f} ms")
