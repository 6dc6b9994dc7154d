"""Dev tool (experiment): the 3x3x3 submanifold conv of U-Net levels 2-5 of the S150k scene under every launch shape the
dispatcher knows (split / non-split / wide, block sizes), back-to-back launches timed with events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoformer_amd import sparse, scene
sc = scene.make_scene(150_000, 1234)
batch = scene.make_batch([sc])
c = batch["voxel_locs"].int().cuda().contiguous()
s = tuple(int(v) for v in batch["spatial_shape"])
M = c.shape[0]
levels = [(sparse.subm_rules(c, sparse.build_index(c, 1, s)), M)]
for L in range(6):
    d = sparse.down_rules(c, 1, s)
    c, s = d.out_coords.contiguous(), d.out_shape
    levels.append((sparse.subm_rules(c, d.index_out), d.M_out))

def timeit(fn, n=30, warm=5):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

NB = 4
for L in (1, 2, 3, 4):
    r, m = levels[L]
    C = 16 * (L + 1)
    R = int((r.nbr[:, :m] >= 0).sum())
    xs = [torch.randn(m, C, device="cuda") for _ in range(NB)]
    res = [torch.randn(m, C, device="cuda") for _ in range(NB)]
    W = torch.randn(27, C, C, device="cuda") * 0.05
    scl = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda")
    byt = 4 * (R * C + 2 * m * C + 27 * C * C) + 8 * R
    ref = None
    for label, kn in (("default", {}), ("split=1", dict(split=1)), ("split=0", dict(split=0)), ("split=1 wide=1", dict(split=1, wide=1)),
                      ("split=0 block=128", dict(split=0, block=128)), ("split=0 block=64", dict(split=0, block=64))):
        sparse.dev_conv_knobs(**kn)
        def run(i):
            return sparse.conv_fwd(xs[i % NB], W, r.nbr, r.gmask, 27, m, r.ld, in_scale=scl, in_shift=sh, residual=res[i % NB])
        try:
            us = timeit(run)
            o = run(0)
            if ref is None: ref = o.clone()
            print(f"level {L + 1} C={C} M={m} R={R} {label:20s} {us:7.2f} us  {byt / us / 1e6:5.2f} TB/s  {2 * R * C * C / us / 1e6:6.1f} TF/s  maxdiff {float((o - ref).abs().max()):.1e}", flush=True)
        except Exception as e:
            print(f"level {L + 1} {label}: {e}")
sparse.dev_conv_knobs()
