"""Dev tool: cycle stamps inside k_ms_hop (library built with -DMS_TRACE: GF_LIB_PATH) for one hop of the search."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene, pointops, _lib
lib = _lib.load()
lib.gf_dev_ms_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
seed, nfg, nq, ms = 1234, 60108, 256, 256
p = scene.make_scene(150_000, seed)["xyz"]
idx = np.sort(np.random.default_rng(seed).permutation(p.shape[0])[:nfg])
xyz = torch.from_numpy(np.ascontiguousarray(p[idx])).cuda()
gd, gi, deg = pointops.knn_radius(xyz, 64, 0.05)
perm = torch.from_numpy(np.random.default_rng(1).permutation(nfg)[:50000]).cuda()
src = pointops.furthest_point_sampling(xyz[perm][None].contiguous(), nq)[0].int().contiguous()
W = (nq + 63) // 64
nw = ((nfg * W + 255) // 256 + 7) // 8 * 8 * 4
for hop in (2, 100, 250):
    buf = torch.zeros(nw * 8, dtype=torch.int64, device="cuda")
    lib.gf_dev_ms_trace(buf.data_ptr(), hop)
    for _ in range(2):
        pointops.geodesic_bfs_ms(gd, gi, src, 0.05, ms)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(nw, 8).astype(np.float64)
    t = t[t[:, 0] > 0]
    t = t[t[:, 6] > 0]
    k0 = t[:, 0].min()
    names = ["start (from the launch's first wave)", "own + parents arrived", "parents' words arrived", "OR + select chain", "edge + dist arrived", "dist stores, later parents", "mask store done"]
    print(f"hop {hop}: {len(t)} waves; kernel span {(t[:, 6].max() - k0) / 100:.2f} us (100 MHz counter)" if False else f"hop {hop}: {len(t)} waves; span first start -> last end {(t[:, 6].max() - k0):.0f} ticks")
    prev = t[:, 0]
    print(f"   wave start after the first wave: mean {np.mean(t[:, 0] - k0):.0f} max {np.max(t[:, 0] - k0):.0f}")
    for i in range(1, 7):
        cur = np.where(t[:, i] > 0, t[:, i], prev)
        d = cur - prev
        print(f"   {names[i]:32s} mean {d.mean():8.0f} p50 {np.median(d):8.0f} p95 {np.percentile(d, 95):8.0f} max {d.max():8.0f}")
        prev = cur
    print(f"   wave lifetime mean {np.mean(prev - t[:, 0]):.0f} max {np.max(prev - t[:, 0]):.0f}")
lib.gf_dev_ms_trace(None, -1)
