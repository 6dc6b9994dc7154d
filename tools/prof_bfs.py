"""Dev tool: only kNN graph + geodesic BFS (256 sources, 60k points), for rocprofv3 --pmc."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import pointops, scene
sc = scene.make_scene(150_000, 1234)
rng = np.random.default_rng(0)
pts = sc["xyz"][rng.permutation(sc["xyz"].shape[0])[:60000]]
xyz = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
D, I, deg = pointops.knn_radius(xyz, 64, 0.05)
src = torch.from_numpy(rng.choice(60000, 256, replace=False).astype(np.int32)).cuda()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    geo = pointops.geodesic_bfs(D, I, deg, src, 0.05, 256)
torch.cuda.synchronize()
print("done", float((geo >= 0).float().mean()))
