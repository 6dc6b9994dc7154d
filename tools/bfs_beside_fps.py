"""Dev tool: the sampling / search stretch of the eval forward in isolation (wall time from the first sampling launch to
the end of both): the two-launch form (first nq picks, then [rest of the sampling || search]) against the gated form
(one sampling launch, the search beside it from the start and waiting inside the kernel), with and without the LDS pad
that keeps search workgroups off the sampler's compute units."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene, pointops, _lib
lib = _lib.load()
main, side = torch.cuda.Stream(), torch.cuda.Stream()
NQ, M = 256, 2048
def two_launch(fps_xyz, gd, gi, deg, wg):
    with torch.cuda.stream(main):
        first = pointops.furthest_point_sampling(fps_xyz, NQ)
        src = first[0, :NQ].contiguous()
        ev = torch.cuda.Event(); ev.record(main)
        idx = pointops.furthest_point_sampling(fps_xyz, M, known=first)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        return pointops.geodesic_bfs(gd, gi, deg, src, 0.05, 256, wg_threads=wg), idx
def gated(fps_xyz, gd, gi, deg, wg, pad):
    with torch.cuda.stream(main):
        idx, gate, ev = pointops.furthest_point_sampling_gated(fps_xyz, M, NQ, lds_pad=pad)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        return pointops.geodesic_bfs_gated(gd, gi, idx[0, :NQ], 0.05, 256, gate, NQ, wg_threads=wg), idx
def wall(fn, reps=8):
    ts = []
    for r in range(reps + 2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return np.mean(ts[2:]) * 1e6, out
for seed, nfg in [(1234, 60108), (1241, 68456), (1250, 90000)]:
    p = scene.make_scene(150_000 if nfg < 80000 else 220_000, seed)["xyz"]
    idx = np.sort(np.random.default_rng(seed).permutation(p.shape[0])[:nfg])
    xyz = torch.from_numpy(np.ascontiguousarray(p[idx])).cuda()
    gd, gi, deg = pointops.knn_radius(xyz, 64, 0.05)
    perm = torch.from_numpy(np.random.default_rng(1).permutation(nfg)[:50000]).cuda()
    fps_xyz = xyz[perm][None].contiguous()
    t0, (g0, i0) = wall(lambda: two_launch(fps_xyz, gd, gi, deg, 512))
    print(f"n {nfg}: two launches, 512 thr {t0:7.1f} us")
    for wg in (512, 1024):
        for pad in (0, 88 * 1024):
            t1, (g1, i1) = wall(lambda: gated(fps_xyz, gd, gi, deg, wg, pad))
            print(f"    gated, {wg} thr, pad {pad // 1024:3d} KB: {t1:7.1f} us  equal {torch.equal(g0, g1) and torch.equal(i0, i1)}")
