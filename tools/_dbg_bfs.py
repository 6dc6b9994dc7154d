import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene, pointops
seed, nfg, nq, ms = 1234, 60108, 256, 40
p = scene.make_scene(150_000, seed)["xyz"]
idx = np.sort(np.random.default_rng(seed).permutation(p.shape[0])[:nfg])
xyz = torch.from_numpy(np.ascontiguousarray(p[idx])).cuda()
gd, gi, deg = pointops.knn_radius(xyz, 64, 0.05)
perm = torch.from_numpy(np.random.default_rng(1).permutation(nfg)[:50000]).cuda()
src = pointops.furthest_point_sampling(xyz[perm][None].contiguous(), nq)[0].int().contiguous()
for ms in (1, 2, 3, 5, 10, 40):
    a = pointops.geodesic_bfs(gd, gi, deg, src, 0.05, ms, wg_threads=512)
    b = pointops.geodesic_bfs_ms(gd, gi, src, 0.05, ms)
    ra, rb = a >= 0, b >= 0
    print("max_step", ms, "reach old", int(ra.sum()), "new", int(rb.sum()), "reach-set diff", int((ra != rb).sum()),
          "value diff among common", int(((a != b) & ra & rb).sum()), "nan in new", int(torch.isnan(b).sum()))
    bad = ((a != b) & ra & rb).nonzero()
    if len(bad):
        q, u = bad[0].tolist()
        print("  first bad q", q, "u", u, "old", a[q, u].item(), "new", b[q, u].item(), "deg", int(deg[u]))
