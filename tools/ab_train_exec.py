"""Dev tool: the batch-4 training step with the native training-mode U-Net (csrc/unet_train.hip) and with the module
tree, alternating inside ONE process (boxes and runs differ by several ms): median ms per step of each.
`ab_train_exec.py lw` alternates the LDS-weight convolution kernel (size-based choice against never) instead."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools import train_dp
from geoformer_amd import parallel, sparse

dev = torch.device("cuda", 0)
args = train_dp.default_args(batch_size=4, fg_frac=0.4)
cfg, m, crit = train_dp.build(args, dev)
red = parallel.BucketedGradReducer(m)
opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, fused=True)
batches = train_dp.make_batches(args, 0, dev, 2)
train_dp.calibrate_foreground(m, batches[0], 0.4)
knob = sys.argv[1] if len(sys.argv) > 1 else "GF_UNET_TRAIN_EXEC"
rounds, per = int(os.environ.get("ROUNDS", 6)), 6
t = {"1": [], "0": []}
n = 0
for r in range(rounds + 1):
    for v in ("1", "0"):
        if knob == "lw":
            sparse.dev_conv_knobs(lw=-1 if v == "1" else 0)
        else:
            os.environ[knob] = v
        for i in range(per):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            train_dp.step(m, crit, red, opt, batches[n % 2], 200, n); n += 1
            torch.cuda.synchronize()
            if r > 0 and i > 0:  # first round and the first step after a switch: warm-up
                t[v].append((time.perf_counter() - t0) * 1e3)
os.environ.pop(knob, None)
for v in ("1", "0"):
    a = np.array(t[v])
    print(f"{knob}={v}: median {np.median(a):.2f} ms  mean {a.mean():.2f}  min {a.min():.2f}  p90 {np.percentile(a, 90):.2f}  n={len(a)}")
