"""Dev tool: only the two level-1 16->16 launches the eval forward issues (first conv of a residual block: activated
input, BatchNorm + ReLU epilogue, dual output; second conv: residual epilogue), for rocprofv3 kernel traces / --pmc."""
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
