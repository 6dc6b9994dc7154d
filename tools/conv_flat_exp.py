"""Dev tool (experiment): the flat-chain kernel (k_conv_flat) against the split / wide shapes on U-Net levels 3-7 of the
S150k scene: 3x3x3 submanifold conv with prologue + residual, the strided conv into the level and the 1x1x1 identity
conv (2C -> C), back-to-back launches timed with events (a dependent chain: each launch reads the previous output)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoformer_amd import sparse, scene
sc = scene.make_scene(150_000, 1234)
batch = scene.make_batch([sc])
c = batch["voxel_locs"].int().cuda().contiguous()
s = tuple(int(v) for v in batch["spatial_shape"])
M = c.shape[0]
levels = [(sparse.subm_rules(c, sparse.build_index(c, 1, s)), M, None)]
for L in range(6):
    d = sparse.down_rules(c, 1, s)
    c, s = d.out_coords.contiguous(), d.out_shape
    levels.append((sparse.subm_rules(c, d.index_out), d.M_out, d))

def timeit(fn, n=40, warm=5):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

SHAPES = (("default(flat=0)", dict(flat=0)), ("flat=1", dict(flat=1)))
for L in (2, 3, 4, 5, 6):
    r, m, d = levels[L]
    C = 16 * (L + 1)
    x0 = torch.randn(m, C, device="cuda"); res = torch.randn(m, C, device="cuda")
    W = torch.randn(27, C, C, device="cuda") * 0.02
    Wi = torch.randn(1, 2 * C, C, device="cuda") * 0.05
    x2 = torch.randn(m, 2 * C, device="cuda")
    scl = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda") * 0.1
    ref = {}
    for label, kn in SHAPES:
        sparse.dev_conv_knobs(**kn)
        state = {"x": x0}
        def run(i):
            # a dependent chain like the forward's: the output of one launch is the next one's input
            state["x"] = sparse.conv_fwd(state["x"], W, r.nbr, r.gmask, 27, m, r.ld, in_scale=scl, in_shift=sh, residual=res)
            return state["x"]
        us = timeit(run)
        state["x"] = x0
        o = run(0)
        us1 = timeit(lambda i: sparse.conv_fwd(x2, Wi, None, None, 1, m, 0))
        o1 = sparse.conv_fwd(x2, Wi, None, None, 1, m, 0)
        if "o" not in ref: ref["o"], ref["o1"] = o.clone(), o1.clone()
        print(f"level {L + 1} C={C} M={m} {label:16s} subm {us:6.2f} us  1x1 {us1:6.2f} us   maxdiff {float((o - ref['o']).abs().max()):.1e} {float((o1 - ref['o1']).abs().max()):.1e}", flush=True)
sparse.dev_conv_knobs()
