"""Dev tool: level-1 (C=16) submanifold conv launch shapes on the S150k scene: time per launch (events over a batch of
launches rotating over several input buffers) and max-abs difference against the first shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import sparse, scene

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sparse.dev_conv_chunks(chunks)
only_p = len(sys.argv) > 3
batch = scene.make_batch([scene.make_scene(150_000, 1234)])
coords = batch["voxel_locs"].int().cuda().contiguous()
shape = tuple(int(s) for s in batch["spatial_shape"])
M = coords.shape[0]
rules = sparse.subm_rules(coords, sparse.build_index(coords, 1, shape))
R = int((rules.nbr[:, :M] >= 0).sum())
print("M", M, "R", R, "steps", None if rules.steps is None else tuple(rules.steps.shape))

def timeit(fn, n=reps, warm=5):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

NB = 6
for Cin in (16, 32):
    xs = [torch.randn(M, Cin, device="cuda") for _ in range(NB)]
    W = torch.randn(27, Cin, 16, device="cuda") * 0.05
    sc = torch.rand(Cin, device="cuda") + 0.5; sh = torch.randn(Cin, device="cuda") * 0.3
    res = [torch.randn(M, 16, device="cuda") for _ in range(NB)]
    outs = [torch.empty(M, 16, device="cuda") for _ in range(NB)]
    byt = 4 * (R * Cin + M * 16 + 27 * Cin * 16) + 8 * R
    osc = torch.rand(16, device="cuda") + 0.5; osh = torch.randn(16, device="cuda") * 0.3
    for name, kw in (("plain", {}), ("aff", dict(in_scale=sc, in_shift=sh)), ("res", dict(residual=True)),
                     ("aff+oact", dict(in_scale=sc, in_shift=sh, out_scale=osc, out_shift=osh)),
                     ("aff+res", dict(in_scale=sc, in_shift=sh, residual=True))):
        ref = None
        for label, knobs in (("pair/os", dict(g16=0)), ("g16 L0 g1", dict(g16=1, g16_ldsw=0, g16_gpw=1, g16_pipe=0)),
                             ("g16 L1 g1", dict(g16=1, g16_ldsw=1, g16_gpw=1, g16_pipe=0)),
                             ("g16p L0", dict(g16=1, g16_ldsw=0, g16_pipe=1)), ("g16p L1", dict(g16=1, g16_ldsw=1, g16_pipe=1))):
            if only_p and "g16p" not in label: continue
            sparse.dev_conv_knobs(**knobs)
            def run(i):
                k = dict(kw)
                if k.get("residual") is True: k["residual"] = res[i % NB]
                return sparse.conv_fwd(xs[i % NB], W, rules.nbr, rules.gmask, 27, M, rules.ld, out=outs[i % NB], steps=rules.steps, **k)
            us = timeit(run)
            o = run(0).clone()
            if ref is None: ref = o
            b = byt + (4 * M * 16 if "res" in name else 0)
            print(f"Cin {Cin} {name:8s} {label:10s} {us:7.2f} us  {b/us/1e6:6.2f} TB/s alg  frac {b/us/1e6/8:.3f}  maxdiff {float((o-ref).abs().max()):.2e}")
sparse.dev_conv_knobs()
