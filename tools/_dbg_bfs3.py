import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene, pointops, _lib
from oracle import oracle
lib = _lib.load()
def pts(n, seed):
    sc = scene.make_scene(max(n, 64), seed)
    p = sc["xyz"]
    return np.ascontiguousarray(p[np.random.default_rng(seed).permutation(p.shape[0])[:n]])
n, nq, max_step = 12000, 256, 256
xyz = pts(n, 5 + n); n = xyz.shape[0]
D2, I = oracle.knn(xyz, xyz, 64); D = np.sqrt(D2)
rng = np.random.default_rng(1)
src = rng.integers(0, n, nq); src[-1] = src[0]
ref = oracle.geodesic(D[:, 1:], I[:, 1:], src, 0.05, max_step)
gd, gi, deg = pointops.knn_radius(torch.from_numpy(xyz).cuda(), 64, 0.05)
s = torch.from_numpy(src.astype(np.int32)).cuda()
old = pointops.geodesic_bfs(gd, gi, deg, s, 0.05, max_step, wg_threads=512).cpu().numpy()
print("old == ref", (old == ref).all())
for t in (0, 1):
    lib.gf_dev_bfs_ms_tiles(t)
    g = pointops.geodesic_bfs_ms(gd, gi, s, 0.05, max_step).cpu().numpy()
    bad = np.argwhere(g != ref)
    print("tiles", t, "equal", (g == ref).all(), "nbad", len(bad))
    for q, u in bad[:6]:
        print("   q", q, "u", u, "src[q]", src[q], "ref", ref[q, u], "got", g[q, u], "dups of src", int((src == src[q]).sum()))
