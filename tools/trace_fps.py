"""Dev tool: phase cycle counts of k_fps (variant built with -DFPS_TRACE; wave 0 of workgroup 0)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene, pointops, _lib
lib = _lib.load()
lib.gf_dev_fps_trace.argtypes = [ctypes.c_void_p]
p = scene.make_scene(150_000, 1234)["xyz"]
idx = np.sort(np.random.default_rng(1).permutation(p.shape[0])[:50000])
xyz = torch.from_numpy(np.ascontiguousarray(p[idx])).cuda()[None].contiguous()
tr = torch.zeros(16, dtype=torch.int64, device="cuda")
lib.gf_dev_fps_trace(tr.data_ptr())
names = ["absorb picks", "wave top-2", "barrier 1", "workgroup merge", "publish + poll", "candidate coordinates", "replay", "barrier 2"]
for m, m0 in ((256, 0), (2048, 0), (2048, 256)):
    first = pointops.furthest_point_sampling(xyz, m0) if m0 else None
    for _ in range(2):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = pointops.furthest_point_sampling(xyz, m, known=first) if m0 else pointops.furthest_point_sampling(xyz, m)
        e.record(); torch.cuda.synchronize()
    t = tr.cpu().numpy().astype(np.float64)
    us = s.elapsed_time(e) * 1e3
    tot = t[:8].sum() - t[3] - t[4] - t[5] - t[6]  # phases 3-6 run inside barrier 2 of the other waves / in wave 0 only
    print(f"picks {m0}..{m}: {us:.0f} us, {int(t[8])} exchanges, {t[9] / max(t[8], 1):.1f} picks per exchange")
    w0 = t[0] + t[1] + t[2] + t[3] + t[4] + t[5] + t[6] + t[7]
    for i, nme in enumerate(names):
        print(f"    {nme:24s} {t[i] / max(t[8], 1):8.0f} ticks per exchange  {100 * t[i] / w0:5.1f} %")
    print(f"    => {us * 1e3 / max(t[8], 1):.0f} ns per exchange, {w0 / max(t[8], 1):.0f} ticks")
