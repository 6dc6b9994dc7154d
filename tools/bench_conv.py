"""Dev tool: time rulebook build + level-1/2 convs on the S150k synthetic scene."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import sparse, scene

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us

sc = scene.make_scene(150_000, 1234)
batch = scene.make_batch([sc])
coords = batch["voxel_locs"].int().cuda().contiguous()
shape = tuple(int(s) for s in batch["spatial_shape"])
M = coords.shape[0]
print("N", sc["xyz"].shape[0], "M", M, "shape", shape)
ix = sparse.build_index(coords, 1, shape)
rules = sparse.subm_rules(coords, ix)
R = int((rules.nbr[:, :M] >= 0).sum())
gm = rules.gmask.cpu().numpy().view(np.uint32)
pop = np.array([bin(int(x)).count("1") for x in gm])
print("R", R, "taps/voxel", R / M, "group offsets avg", pop.mean())
print("index build us", timeit(lambda: sparse.build_index(coords, 1, shape)))
print("subm rules us", timeit(lambda: sparse.subm_rules(coords, ix)))
print("down rules us", timeit(lambda: sparse.down_rules(coords, 1, shape)))
levels = [(coords, shape, rules, M)]
c, s = coords, shape
for L in range(6):
    d = sparse.down_rules(c, 1, s)
    c, s = d.out_coords.contiguous(), d.out_shape
    r = sparse.subm_rules(c, d.index_out)
    levels.append((c, s, r, d.M_out))
for L, (c, s, r, m) in enumerate(levels):
    C = 16 * (L + 1)
    x = torch.randn(m, C, device="cuda")
    W = torch.randn(27, C, C, device="cuda") * 0.05
    Rl = int((r.nbr[:, :m] >= 0).sum())
    us = timeit(lambda: sparse.conv_fwd(x, W, r.nbr, r.gmask, 27, m, r.ld))
    byt = 4 * (Rl * C + m * C + 27 * C * C) + 8 * Rl
    fl = 2 * Rl * C * C
    print(f"level {L+1} M {m} C {C} R {Rl}: {us:.1f} us  alg {byt/us/1e6:.2f} TB/s  {fl/us/1e6:.2f} TFLOP/s")
# fused-variant experiment on level 1
c, s, r, m = levels[0]
x = torch.randn(m, 16, device="cuda"); W = torch.randn(27, 16, 16, device="cuda") * 0.05
sc = torch.rand(16, device="cuda") + 0.5; sh = torch.randn(16, device="cuda"); res = torch.randn(m, 16, device="cuda")
for name, kw in (("plain", {}), ("scale", dict(in_scale=sc, in_shift=sh)), ("resid", dict(residual=res)),
                 ("both", dict(in_scale=sc, in_shift=sh, residual=res))):
    print("L1", name, round(timeit(lambda: sparse.conv_fwd(x, W, r.nbr, r.gmask, 27, m, r.ld, **kw)), 1), "us")
xr = torch.relu(x)
print("L1 plain on relu'd input", round(timeit(lambda: sparse.conv_fwd(xr, W, r.nbr, r.gmask, 27, m, r.ld)), 1), "us")
