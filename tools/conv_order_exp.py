"""Dev tool: how much does the ROW ORDER of the level-1 voxels matter for the submanifold conv?  The same problem with
its rows in the scene's order, in raster order (ascending linear index), in Morton order and shuffled: mean number of
kernel offsets per 32-row pair (the union mask the paired kernel walks) and the launch time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import sparse, scene


def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def part1by2(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


sc = scene.make_scene(150_000, 1234)
batch = scene.make_batch([sc])
coords = batch["voxel_locs"].int().cuda().contiguous()
shape = tuple(int(s) for s in batch["spatial_shape"])
M = coords.shape[0]
c = coords.long()
lin = ((c[:, 0] * shape[0] + c[:, 1]) * shape[1] + c[:, 2]) * shape[2] + c[:, 3]
morton = part1by2(c[:, 1]) | (part1by2(c[:, 2]) << 1) | (part1by2(c[:, 3]) << 2)
blk = 4  # 4x4x4 bricks in raster order, raster inside
brick = (((c[:, 1] // blk) * 4096 + (c[:, 2] // blk)) * 4096 + (c[:, 3] // blk)) * 64 + ((c[:, 1] % blk) * 16 + (c[:, 2] % blk) * 4 + c[:, 3] % blk)
orders = {"scene order": torch.arange(M, device="cuda"), "raster": torch.argsort(lin), "morton": torch.argsort(morton),
          "bricks4": torch.argsort(brick), "shuffled": torch.randperm(M, device="cuda")}
x = torch.randn(M, 16, device="cuda"); W = torch.randn(27, 16, 16, device="cuda") * 0.05
sc_ = torch.rand(16, device="cuda") + 0.5; sh_ = torch.randn(16, device="cuda"); res = torch.randn(M, 16, device="cuda")
for name, p in orders.items():
    cp = coords[p].contiguous()
    r = sparse.subm_rules(cp, sparse.build_index(cp, 1, shape))
    gm = r.gmask[: (M + 15) // 16].cpu().numpy().view(np.uint32)
    pairs = gm[: (len(gm) // 2) * 2].reshape(-1, 2)
    w1 = np.mean([bin(int(v)).count("1") for v in gm])
    w2 = np.mean([bin(int(a | b)).count("1") for a, b in pairs])
    xp = x[p].contiguous(); rp = res[p].contiguous()
    t_plain = timeit(lambda: sparse.conv_fwd(xp, W, r.nbr, r.gmask, 27, M, r.ld))
    t_full = timeit(lambda: sparse.conv_fwd(xp, W, r.nbr, r.gmask, 27, M, r.ld, in_scale=sc_, in_shift=sh_, residual=rp))
    print(f"{name:12s} offsets per 16-row group {w1:5.2f}  per 32-row pair {w2:5.2f}   plain {t_plain:6.2f} us   bn+relu+residual {t_full:6.2f} us")
