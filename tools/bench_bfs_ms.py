"""Dev tool: the multi-source BFS (csrc/geodesic_ms.hip) against the per-query LDS kernel on S150k-like foregrounds:
bit equality, time alone, the hop count, and per-kernel times of the build (reverse CSR) and the search."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene, pointops, _lib
lib = _lib.load()


def timeit(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): out = fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3, out


cases = [(1234, 60108, 256, 256), (1241, 68456, 256, 256), (1239, 71016, 128, 128), (1250, 109000, 256, 256)]
if len(sys.argv) > 1 and sys.argv[1] == "small":
    cases = [(7, 6000, 40, 64)]
for seed, nfg, nq, ms in cases:
    p = scene.make_scene(150_000 if nfg < 100000 else 250_000, seed)["xyz"]
    idx = np.sort(np.random.default_rng(seed).permutation(p.shape[0])[:nfg])
    xyz = torch.from_numpy(np.ascontiguousarray(p[idx])).cuda()
    gd, gi, deg = pointops.knn_radius(xyz, 64, 0.05)
    perm = torch.from_numpy(np.random.default_rng(1).permutation(nfg)[:min(50000, nfg)]).cuda()
    src = pointops.furthest_point_sampling(xyz[perm][None].contiguous(), nq)[0].int().contiguous()
    t_old, g_old = timeit(lambda: pointops.geodesic_bfs(gd, gi, deg, src, 0.05, ms, wg_threads=512))
    t_old1k, _ = timeit(lambda: pointops.geodesic_bfs(gd, gi, deg, src, 0.05, ms, wg_threads=1024))
    t_new, g_new = timeit(lambda: pointops.geodesic_bfs_ms(gd, gi, src, 0.05, ms))
    t_til, g_til = timeit(lambda: pointops.geodesic_bfs_ms(gd, gi, src, 0.05, ms, xyz=xyz))
    lib.gf_dev_bfs_ms_persist(1)
    t_per, (g_per, flag) = timeit(lambda: pointops.geodesic_bfs_ms(gd, gi, src, 0.05, ms, xyz=xyz, return_flag=True))
    lib.gf_dev_bfs_ms_persist(-1)
    print("   one launch (persistent tiles):", round(t_per, 1), "us, timeout flag", int(flag.item()), "equal", torch.equal(g_old, g_per))
    eq = torch.equal(g_old, g_new) and torch.equal(g_old, g_til)
    hops = int((g_new >= 0).any(0).sum().item())
    print(f"seed {seed} n {nfg} nq {nq} max_step {ms}: per-query 512thr {t_old:8.1f} us, 1024thr {t_old1k:8.1f} us | multi-source gather {t_new:8.1f} us, spatial tiles {t_til:8.1f} us"
          f" | equal {eq} | mean degree {deg.float().mean().item():.1f} reached/query {(g_new >= 0).sum(1).float().mean().item():.0f}")
    if not eq:
        bad = (g_old != g_new)
        print("   mismatches:", int(bad.sum().item()), "first:", bad.nonzero()[:5].tolist(),
              g_old[bad][:5].tolist(), g_new[bad][:5].tolist())
