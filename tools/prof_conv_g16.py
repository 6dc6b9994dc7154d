"""Dev tool: launch only the level-1 16->16 conv (plain and aff+res) a number of times, for rocprofv3 kernel traces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoformer_amd import sparse, scene
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ldsw = int(sys.argv[2]) if len(sys.argv) > 2 else 1
gpw = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pipe = int(sys.argv[4]) if len(sys.argv) > 4 else 1
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
sc = torch.rand(16, device="cuda") + 0.5; sh = torch.randn(16, device="cuda") * 0.3
sparse.dev_conv_knobs(g16=1, g16_ldsw=ldsw, g16_gpw=gpw, g16_pipe=pipe)
for i in range(reps):
    sparse.conv_fwd(xs[i % NB], W, rules.nbr, rules.gmask, 27, M, rules.ld, out=outs[i % NB], steps=rules.steps)
for i in range(reps):
    sparse.conv_fwd(xs[i % NB], W, rules.nbr, rules.gmask, 27, M, rules.ld, out=outs[i % NB], steps=rules.steps,
                    in_scale=sc, in_shift=sh, residual=res[i % NB])
torch.cuda.synchronize()
