"""Dev tool: geodesic BFS on the S150k foreground (60k points, 256 queries): pull-resolved vs bidding kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene, pointops, _lib
lib = _lib.load()
for seed, nfg in ((1234, 60108), (1241, 68456)):
    p = scene.make_scene(150_000, seed)["xyz"]
    idx = np.sort(np.random.default_rng(seed).permutation(p.shape[0])[:nfg])
    xyz = torch.from_numpy(np.ascontiguousarray(p[idx])).cuda()
    gd, gi, deg = pointops.knn_radius(xyz, 64, 0.05)
    src = torch.from_numpy(np.random.default_rng(1).integers(0, nfg, 256).astype(np.int32)).cuda()
    outs = {}
    for pull in (0,):
        for wg in (256, 1024):
            for _ in range(2): geo = pointops.geodesic_bfs(gd, gi, deg, src, 0.05, 256, wg_threads=wg)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): geo = pointops.geodesic_bfs(gd, gi, deg, src, 0.05, 256, wg_threads=wg)
            e.record(); torch.cuda.synchronize()
            outs[(pull, wg)] = geo
            print(f"seed {seed} n {nfg} pull {pull} wg {wg}: {s.elapsed_time(e)/5*1e3:8.1f} us  reached/query {(geo>=0).sum(1).float().mean().item():.0f} maxgeo {geo.max().item():.2f}")
    assert all(torch.equal(outs[(0, 256)], v) for v in outs.values())
