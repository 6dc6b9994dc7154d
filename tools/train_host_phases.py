"""Dev tool: where the HOST spends a batch-4 training step (tools/train_dp.py's step, calibrated foreground): host
timestamps at the phase boundaries without extra synchronisation, then cProfile of three steps (CPROFILE=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools import train_dp
from geoformer_amd import parallel

dev = torch.device("cuda", 0)
args = train_dp.default_args(batch_size=4, fg_frac=0.4)
cfg, m, crit = train_dp.build(args, dev)
red = parallel.BucketedGradReducer(m)
opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, fused=True)
batches = train_dp.make_batches(args, 0, dev, 2)
train_dp.calibrate_foreground(m, batches[0], 0.4)
for i in range(4):
    train_dp.step(m, crit, red, opt, batches[i % 2], 200, i)
torch.cuda.synchronize()
acc = np.zeros(6)
n = 8
for i in range(n):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    np.random.seed(i)
    red.prepare()
    out = m(batches[i % 2], 200); t.append(time.perf_counter())
    loss, info = crit(out, batches[i % 2], 200); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    red.finish(); opt.step(); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    acc[:5] += np.diff(t)
    acc[5] += t[-1] - t[0]
print("host ms: forward %.1f  criterion %.1f  backward %.1f  optimizer %.1f  device tail %.1f  total %.1f" % tuple(acc / n * 1e3))
if os.environ.get("CPROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(3):
        train_dp.step(m, crit, red, opt, batches[i % 2], 200, 50 + i)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(35)
if os.environ.get("BWD_CLASSES"):
    import collections
    acc2 = collections.defaultdict(lambda: [0, 0.0])
    def subclasses(c):
        for s in c.__subclasses__():
            yield s
            yield from subclasses(s)
    for cls in set(subclasses(torch.autograd.Function)):
        if not cls.__module__.startswith(("geoformer_amd", "tools", "__main__")):
            continue
        orig = cls.backward
        def make(orig, name):
            def wrapped(ctx, *a):
                t0 = time.perf_counter()
                r = orig(ctx, *a)
                e = acc2[name]; e[0] += 1; e[1] += time.perf_counter() - t0
                return r
            return wrapped
        cls.backward = staticmethod(make(orig, cls.__module__.split(".")[-1] + "." + cls.__name__))
    k = 4
    tb = 0.0
    for i in range(k):
        np.random.seed(i); red.prepare()
        out = m(batches[i % 2], 200); loss, info = crit(out, batches[i % 2], 200)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss.backward(); tb += time.perf_counter() - t0
        red.finish(); opt.step()
    torch.cuda.synchronize()
    print("host backward (device idle at its start) %.1f ms per step; Python backward functions:" % (tb / k * 1e3))
    tot = 0.0
    for name, (c, t) in sorted(acc2.items(), key=lambda kv: -kv[1][1]):
        print(f"  {name:45s} calls/step {c / k:6.1f}  ms/step {t / k * 1e3:6.2f}  us/call {t / c * 1e6:6.1f}")
        tot += t
    print("  total %.2f ms per step" % (tot / k * 1e3))
if os.environ.get("GRAPH_NODES"):
    import collections
    np.random.seed(0); red.prepare()
    out = m(batches[0], 200); loss, info = crit(out, batches[0], 200)
    seen, stack, hist = set(), [loss.grad_fn], collections.Counter()
    while stack:
        f = stack.pop()
        if f is None or f in seen:
            continue
        seen.add(f)
        hist[type(f).__name__] += 1
        stack.extend(g for g, _ in f.next_functions)
    print("autograd nodes:", len(seen))
    for k, v in hist.most_common(40):
        print(f"  {k:40s} {v}")
    loss.backward(); red.finish()
