"""Dev tool: geodesic BFS on an S150k-like foreground (60-68k points, 256 queries): workgroup sizes, the
hash-resolved kernel (GF_BFS_HASH unset) against the global-key kernel (GF_BFS_HASH=0 in the environment), the number
of queries the hash kernel handed to the fallback, and equality of the results across workgroup sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene, pointops, _lib
lib = _lib.load()
print("GF_BFS_HASH =", os.environ.get("GF_BFS_HASH"))
for seed, nfg in ((1234, 60108), (1241, 68456)):
    p = scene.make_scene(150_000, seed)["xyz"]
    idx = np.sort(np.random.default_rng(seed).permutation(p.shape[0])[:nfg])
    xyz = torch.from_numpy(np.ascontiguousarray(p[idx])).cuda()
    gd, gi, deg = pointops.knn_radius(xyz, 64, 0.05)
    src = torch.from_numpy(np.random.default_rng(1).integers(0, nfg, 256).astype(np.int32)).cuda()
    outs = {}
    for wg in (256, 512, 1024):
        for _ in range(2): geo = pointops.geodesic_bfs(gd, gi, deg, src, 0.05, 256, wg_threads=wg)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): geo = pointops.geodesic_bfs(gd, gi, deg, src, 0.05, 256, wg_threads=wg)
        e.record(); torch.cuda.synchronize()
        outs[wg] = geo
        print(f"seed {seed} n {nfg} wg {wg}: {s.elapsed_time(e)/5*1e3:8.1f} us  reached/query {(geo>=0).sum(1).float().mean().item():.0f} maxgeo {geo.max().item():.2f}")
    assert all(torch.equal(outs[256], v) for v in outs.values())
    torch.save(outs[256].cpu(), f"/tmp/bfs_{seed}_{os.environ.get('GF_BFS_HASH', '1')}.pt")
    other = f"/tmp/bfs_{seed}_{'0' if os.environ.get('GF_BFS_HASH', '1') != '0' else '1'}.pt"
    if os.path.exists(other):
        print("equal to the other variant's result:", torch.equal(torch.load(other), outs[256].cpu()))
