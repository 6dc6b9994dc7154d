"""Dev tool: only the level-1 16->16 submanifold conv of the S150k scene, a few launches (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoformer_amd import sparse, scene
sc = scene.make_scene(150_000, 1234)
batch = scene.make_batch([sc])
coords = batch["voxel_locs"].int().cuda().contiguous()
shape = tuple(int(s) for s in batch["spatial_shape"])
M = coords.shape[0]
rules = sparse.subm_rules(coords, sparse.build_index(coords, 1, shape))
x = torch.randn(M, 16, device="cuda"); W = torch.randn(27, 16, 16, device="cuda") * 0.05
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    y = sparse.conv_fwd(x, W, rules.nbr, rules.gmask, 27, M, rules.ld)
torch.cuda.synchronize()
print("done", float(y.abs().mean()))
