import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from geoformer_amd import sparse, scene
sc = scene.make_scene(150_000, 1234)
batch = scene.make_batch([sc])
coords = batch["voxel_locs"].int().cuda().contiguous()
shape = tuple(int(s) for s in batch["spatial_shape"])
M = coords.shape[0]
rules = sparse.subm_rules(coords, sparse.build_index(coords, 1, shape))
x = torch.randn(M, 16, device="cuda"); W = torch.randn(27, 16, 16, device="cuda") * 0.05
def timeit(fn, n=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
full = rules.gmask
for name, keep in (("none", 0), ("centre only", 1 << 13), ("3 offsets", (1 << 13) | (1 << 12) | (1 << 14)),
                   ("9 in-plane (dz=0.. k%3==1)", sum(1 << k for k in range(27) if k % 3 == 1)), ("all", (1 << 27) - 1)):
    gm = (full & keep).contiguous()
    pop = np.array([bin(int(v) & 0xffffffff).count("1") for v in gm.cpu().numpy()]).mean()
    print(f"{name:28s} mean steps {pop:5.2f}  {timeit(lambda: sparse.conv_fwd(x, W, rules.nbr, gm, 27, M, rules.ld)):6.1f} us")
y = torch.empty(M, 16, device="cuda")
print("copy M x 16 fp32", round(timeit(lambda: y.copy_(x)), 1), "us")
gm0 = (full & 0).contiguous()
for frac in (0.125, 0.25, 0.5, 1.0):
    m = int(M * frac) // 16 * 16
    print("none-mask, M_out", m, round(timeit(lambda: sparse.conv_fwd(x, W, rules.nbr, gm0, 27, m, rules.ld)), 1), "us")
for frac in (0.125, 0.25, 0.5, 1.0):
    m = int(M * frac) // 16 * 16
    print("full-mask, M_out", m, round(timeit(lambda: sparse.conv_fwd(x, W, rules.nbr, full, 27, m, rules.ld)), 1), "us")
z = torch.zeros(M, 16, device="cuda")
print("zero_ M x16", round(timeit(lambda: z.zero_()), 1), "us")
