"""Dev tool: the BFS alone with the forward's kind of sources (the first 256 furthest-point-sampling picks of a random
50 000-point subset, indices applied to the un-permuted points like the reference does) against random sources."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene, pointops
for seed, nfg in ((1234, 60108), (1241, 68456), (1239, 71016)):
    p = scene.make_scene(150_000, seed)["xyz"]
    idx = np.sort(np.random.default_rng(seed).permutation(p.shape[0])[:nfg])
    xyz = torch.from_numpy(np.ascontiguousarray(p[idx])).cuda()
    gd, gi, deg = pointops.knn_radius(xyz, 64, 0.05)
    perm = torch.from_numpy(np.random.default_rng(1).permutation(nfg)[:50000]).cuda()
    picks = pointops.furthest_point_sampling(xyz[perm][None].contiguous(), 256)[0]
    srcs = {"random": torch.from_numpy(np.random.default_rng(1).integers(0, nfg, 256).astype(np.int32)).cuda(),
            "fps picks": picks.int().contiguous()}
    for name, src in srcs.items():
        for wg in (256, 512, 1024):
            for _ in range(2): geo = pointops.geodesic_bfs(gd, gi, deg, src, 0.05, 256, wg_threads=wg)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): geo = pointops.geodesic_bfs(gd, gi, deg, src, 0.05, 256, wg_threads=wg)
            e.record(); torch.cuda.synchronize()
            print(f"seed {seed} n {nfg} {name:10s} wg {wg}: {s.elapsed_time(e)/5*1e3:8.1f} us  reached/query {(geo>=0).sum(1).float().mean().item():.0f} maxgeo {geo.max().item():.2f}")
