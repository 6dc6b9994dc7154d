"""Dev tool: one scene's multi-source BFS under rocprofv3 (kernel trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene, pointops
seed, nfg, nq, ms = 1234, 60108, 256, 256
p = scene.make_scene(150_000, seed)["xyz"]
idx = np.sort(np.random.default_rng(seed).permutation(p.shape[0])[:nfg])
xyz = torch.from_numpy(np.ascontiguousarray(p[idx])).cuda()
gd, gi, deg = pointops.knn_radius(xyz, 64, 0.05)
perm = torch.from_numpy(np.random.default_rng(1).permutation(nfg)[:50000]).cuda()
src = pointops.furthest_point_sampling(xyz[perm][None].contiguous(), nq)[0].int().contiguous()
for _ in range(3):
    g = pointops.geodesic_bfs_ms(gd, gi, src, 0.05, ms, xyz=xyz)
torch.cuda.synchronize()
