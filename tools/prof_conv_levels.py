"""Dev tool: the submanifold conv of every U-Net level of the S150k scene (BN+ReLU prologue, residual epilogue
like the eval backbone), a few launches each -- for rocprofv3 --kernel-trace (true kernel durations)."""
import sys, os
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for L, (r, m) in enumerate(levels):
    C = 16 * (L + 1)
    x = torch.randn(m, C, device="cuda"); W = torch.randn(27, C, C, device="cuda") * 0.05
    scl = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda"); res = torch.randn(m, C, device="cuda")
    for _ in range(n):
        y = sparse.conv_fwd(x, W, r.nbr, r.gmask, 27, m, r.ld, in_scale=scl, in_shift=sh, residual=res)
torch.cuda.synchronize()
print("done")
