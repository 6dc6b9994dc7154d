"""Dev tool (experiment): waves per workgroup of the pipelined level-1 conv kernel (weights staged once per workgroup)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import sparse, scene

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
batch = scene.make_batch([scene.make_scene(150_000, 1234)])
coords = batch["voxel_locs"].int().cuda().contiguous()
shape = tuple(int(s) for s in batch["spatial_shape"])
M = coords.shape[0]

def timeit(fn, n=reps, warm=5):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

NB = 6
xs = [torch.randn(M, 16, device="cuda") for _ in range(NB)]
W = torch.randn(27, 16, 16, device="cuda") * 0.05
res = [torch.randn(M, 16, device="cuda") for _ in range(NB)]
outs = [torch.empty(M, 16, device="cuda") for _ in range(NB)]
osc = torch.rand(16, device="cuda") + 0.5; osh = torch.randn(16, device="cuda") * 0.3
ref = None
for chunks in (3072, 2048, 4096):
    sparse.dev_conv_chunks(chunks)
    rules = sparse.subm_rules(coords, sparse.build_index(coords, 1, shape))
    R = int((rules.nbr[:, :M] >= 0).sum())
    byt = 4 * (R * 16 + M * 16 + 27 * 256) + 8 * R
    for wpb in (4, 8, 12, 16):
        if chunks % wpb: continue
        sparse.dev_conv_g16p_wpb(wpb)
        for name, kw in (("res", dict(residual=True)), ("oact", dict(out_scale=osc, out_shift=osh))):
            def run(i):
                k = dict(kw)
                if k.get("residual") is True: k["residual"] = res[i % NB]
                return sparse.conv_fwd(xs[i % NB], W, rules.nbr, rules.gmask, 27, M, rules.ld, out=outs[i % NB], steps=rules.steps, **k)
            us = timeit(run)
            o = run(0).clone()
            if name == "res":
                if ref is None: ref = o
                d = float((o - ref).abs().max())
            b = byt + (4 * M * 16 if name == "res" else 0)
            print(f"chunks {chunks} wpb {wpb:2d} {name:5s} {us:6.2f} us  frac {b / us / 1e6 / 8:.3f}  maxdiff {d:.1e}", flush=True)
sparse.dev_conv_g16p_wpb(0); sparse.dev_conv_chunks(0)
