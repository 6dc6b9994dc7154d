"""Dev tool: phase cycle totals of k_ms_persist (library built with -DMS_TRACE: GF_LIB_PATH)."""
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
tiles = (nfg + 255) // 256
buf = torch.zeros(tiles * 16 * 4, dtype=torch.int64, device="cuda")
lib.gf_dev_ms_trace(buf.data_ptr(), -1)
lib.gf_dev_bfs_ms_persist(1)
for _ in range(2):
    g = pointops.geodesic_bfs_ms(gd, gi, src, 0.05, ms, xyz=xyz)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(tiles, 16, 4).astype(np.float64) / ms
names = ["wait for the neighbours' counters (+ barrier)", "pull the halo rows (+ barrier)", "OR, parents, distances", "barrier, publish, drain, barrier, counter"]
print("cycles per hop, mean over tiles; wave 0 / mean of the other waves / max over tiles (wave 0)")
for i, nme in enumerate(names):
    print(f"   {nme:48s} {t[:, 0, i].mean():8.0f} {t[:, 1:, i].mean():8.0f} {t[:, 0, i].max():8.0f}")
print("   total per hop (wave 0):", t[:, 0, :].sum(1).mean())
