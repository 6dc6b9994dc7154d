"""Dev tool: the pipelined-distance BFS kernel (k_geodesic_bfs_pipe) against k_geodesic_bfs_lds on S150k-like
foregrounds: bit-equality of the distances and time per launch at 512 / 1024 threads per query (alone on the device)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene, pointops, _lib
lib = _lib.load()
for seed, nfg, nq, ms in ((77, 150000, 128, 128), (1241, 68456, 256, 256), (1234, 60108, 256, 256)):
    npts = 150_000 if nfg < 100000 else 250_000
    p = scene.make_scene(npts, seed)["xyz"]
    idx = np.sort(np.random.default_rng(seed).permutation(p.shape[0])[:nfg])
    xyz = torch.from_numpy(np.ascontiguousarray(p[idx])).cuda()
    gd, gi, deg = pointops.knn_radius(xyz, 64, 0.05)
    src = torch.from_numpy(np.random.default_rng(1).integers(0, idx.shape[0], nq).astype(np.int32)).cuda()
    res = {}
    for pipe in (0, 1):
        lib.gf_dev_bfs_pipe(pipe)
        for wg in (512, 1024):
            for _ in range(2): geo = pointops.geodesic_bfs(gd, gi, deg, src, 0.05, ms, wg_threads=wg)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): geo = pointops.geodesic_bfs(gd, gi, deg, src, 0.05, ms, wg_threads=wg)
            e.record(); torch.cuda.synchronize()
            res[(pipe, wg)] = geo
            hops = "-"
            print(f"seed {seed} n {idx.shape[0]} nq {nq} pipe {pipe} wg {wg}: {s.elapsed_time(e)/5*1e3:8.1f} us  reached/query {(geo>=0).sum(1).float().mean().item():.0f} maxgeo {geo.max().item():.3f}", flush=True)
    lib.gf_dev_bfs_pipe(-1)
    ref = res[(0, 512)]
    for k, v in res.items():
        eq = torch.equal(ref, v)
        print("   ", k, "equal to lds@512:", eq, "" if eq else f"mismatches {(ref != v).sum().item()} maxdiff {(ref - v).abs().max().item()}")
if os.environ.get("BFS_PIPE_TRACE"):
    # a -DBFS_PIPE_TRACE build (GF_LIB_PATH): thread 0's cycle stamps per query, summed over the hops
    from geoformer_amd._lib import ptr, stream_ptr, check
    names = ["expansion", "drain", "barrier A", "mark", "barrier B", "hops", "ring sum", "m:visited", "m:prefetch", "m:S3", "m:S2", "m:S1"]
    for wg in (512, 1024):
        n, K = gd.shape
        geo = torch.empty((nq, n), dtype=torch.float32, device="cuda")
        keys = torch.empty((nq, n), dtype=torch.int64, device="cuda")
        queues = torch.zeros((nq, 10 * n), dtype=torch.int32, device="cuda")
        lib.gf_dev_bfs_pipe(1)
        check(lib.gf_geodesic_bfs_cfg(ptr(gd), ptr(gi), ptr(deg), n, K, ptr(src), nq, 0.05, ms, ptr(geo), ptr(keys), ptr(queues), wg, stream_ptr()), "bfs")
        torch.cuda.synchronize()
        tr = queues.view(torch.int64).view(nq, 5 * n)[:, 2 * n:2 * n + 12].cpu().numpy().astype(np.float64)
        hops = tr[:, 5].mean()
        print(f"trace wg {wg}: hops {hops:.0f} ring {tr[:, 6].sum() / tr[:, 5].sum():.0f} | per hop ticks: " +
              ", ".join(f"{names[i]} {tr[:, i].sum() / tr[:, 5].sum():.0f}" for i in (0, 1, 2, 3, 4, 7, 8, 9, 10, 11)))
