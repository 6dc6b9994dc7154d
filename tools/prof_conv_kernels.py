"""Dev tool: the conv launches whose counters are quoted (for rocprofv3 kernel traces / --pmc): the two level-1 16->16
launches the eval forward issues (k_conv_g16p: residual epilogue / BatchNorm + ReLU epilogue), and the LDS-weight kernel's
three launch shapes (k_conv_lw: level-2 32->32 with residual and with prologue + epilogue, level-1 32->16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoformer_amd import sparse, scene
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = scene.make_batch([scene.make_scene(150_000, 1234)])
coords = batch["voxel_locs"].int().cuda().contiguous()
shape = tuple(int(s) for s in batch["spatial_shape"])
M = coords.shape[0]
rules = sparse.subm_rules(coords, sparse.build_index(coords, 1, shape))
NB = 6
xs = [torch.randn(M, 16, device="cuda") for _ in range(NB)]
res = [torch.randn(M, 16, device="cuda") for _ in range(NB)]
outs = [torch.empty(M, 16, device="cuda") for _ in range(NB)]
W = torch.randn(27, 16, 16, device="cuda") * 0.05
osc = torch.rand(16, device="cuda") + 0.5; osh = torch.randn(16, device="cuda") * 0.3
for i in range(reps):  # second conv of a block: residual epilogue
    sparse.conv_fwd(xs[i % NB], W, rules.nbr, rules.gmask, 27, M, rules.ld, out=outs[i % NB], steps=rules.steps,
                    residual=res[i % NB])
for i in range(reps):  # first conv of a block: the consumer's BatchNorm + ReLU in the epilogue
    sparse.conv_fwd(xs[i % NB], W, rules.nbr, rules.gmask, 27, M, rules.ld, out=outs[i % NB], steps=rules.steps,
                    out_scale=osc, out_shift=osh)
torch.cuda.synchronize()

# ---- k_conv_lw: level 1 (32 -> 16) and level 2 (32 -> 32) ----
x32 = [torch.randn(M, 32, device="cuda") for _ in range(NB)]
W3216 = torch.randn(27, 32, 16, device="cuda") * 0.05
sc32 = torch.rand(32, device="cuda") + 0.5; sh32 = torch.randn(32, device="cuda") * 0.3
sparse.FLAT_MIN_ROWS = 0
flat1 = sparse.flat_steps(rules.nbr, rules.gmask, 27, M, rules.ld)
for i in range(reps):
    sparse.conv_fwd(x32[i % NB], W3216, rules.nbr, rules.gmask, 27, M, rules.ld, out=outs[i % NB], steps=rules.steps, flat=flat1,
                    in_scale=sc32, in_shift=sh32, out_scale=osc, out_shift=osh)
d = sparse.down_rules(coords, 1, shape)
c2, shp2 = d.out_coords.contiguous(), d.out_shape
M2 = c2.shape[0]
r2 = sparse.subm_rules(c2, sparse.build_index(c2, 1, shp2))
y32 = [torch.randn(M2, 32, device="cuda") for _ in range(NB)]
o32 = [torch.empty(M2, 32, device="cuda") for _ in range(NB)]
r32 = [torch.randn(M2, 32, device="cuda") for _ in range(NB)]
W3232 = torch.randn(27, 32, 32, device="cuda") * 0.05
for i in range(reps):
    sparse.conv_fwd(y32[i % NB], W3232, r2.nbr, r2.gmask, 27, M2, r2.ld, out=o32[i % NB], flat=r2.flat, residual=r32[i % NB])
for i in range(reps):
    sparse.conv_fwd(y32[i % NB], W3232, r2.nbr, r2.gmask, 27, M2, r2.ld, out=o32[i % NB], flat=r2.flat, in_scale=sc32,
                    in_shift=sh32, out_scale=sc32, out_shift=sh32)
torch.cuda.synchronize()
