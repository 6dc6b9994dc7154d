import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene, pointops, _lib
lib = _lib.load()
def pts(n, seed):
    sc = scene.make_scene(max(n, 64), seed)
    p = sc["xyz"]
    return np.ascontiguousarray(p[np.random.default_rng(seed).permutation(p.shape[0])[:n]])
for n, nq in ((12000, 256), (12000, 128), (30000, 256), (60000, 256)):
    xyz = torch.from_numpy(pts(n, 5 + n)).cuda()
    n = xyz.shape[0]
    gd, gi, deg = pointops.knn_radius(xyz, 64, 0.05)
    src = torch.from_numpy(np.random.default_rng(1).integers(0, n, nq).astype(np.int32)).cuda()
    for ms in (1, 2, 3, 8, 256):
        lib.gf_dev_bfs_ms_tiles(0); a = pointops.geodesic_bfs_ms(gd, gi, src, 0.05, ms).clone()
        lib.gf_dev_bfs_ms_tiles(1); b = pointops.geodesic_bfs_ms(gd, gi, src, 0.05, ms).clone()
        ra, rb = a >= 0, b >= 0
        bad = (ra != rb).nonzero()
        print("n", n, "nq", nq, "max_step", ms, "deg mean", float(deg.float().mean()), "reach gather", int(ra.sum()), "tiles", int(rb.sum()), "diff", len(bad), "val diff", int(((a != b) & ra & rb).sum()))
        if len(bad):
            q, u = bad[0].tolist(); print("   first q", q, "u", u, "tile", u // 256, "lu", u % 256, "gather", a[q, u].item(), "tiles", b[q, u].item())
            break
lib.gf_dev_bfs_ms_tiles(-1)
