"""Dev tool: cycle stamps of the sampling kernel's wave 0 (workgroup 0) per phase, on a -DFPS_TRACE build:
   tools/build_variant.sh fps_trace pointops.hip -DFPS_TRACE;  GF_LIB_PATH=geoformer_amd/lib/exp/fps_trace.so python tools/trace_fps.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene, pointops, _lib
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
tr = torch.zeros(10, dtype=torch.int64, device="cuda")
raw.gf_dev_fps_trace.argtypes = [ctypes.c_void_p]
assert raw.gf_dev_fps_trace(tr.data_ptr()) == 0
p = scene.make_scene(150_000, 1234)["xyz"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
idx = np.random.default_rng(1).permutation(p.shape[0])[:n]
xyz = torch.from_numpy(np.ascontiguousarray(p[idx])).cuda()[None].contiguous()
for _ in range(2): pointops.furthest_point_sampling(xyz, 2048)
torch.cuda.synchronize()
t = tr.cpu().numpy()
names = ["lane best (+absorb)", "wave top-2", "barrier A", "wg merge", "publish + poll", "cand coords", "replay", "barrier B"]
rounds, picks = int(t[8]), int(t[9])
tot = sum(int(x) for x in t[:8])
print(f"rounds {rounds} picks {picks}  stamped cycles {tot} = {tot / 100e6 * 1e3:.3f} ms at 100 MHz (s_memtime)")
for i, nm in enumerate(names):
    print(f"  {nm:22s} {int(t[i]):10d}  {100.0 * int(t[i]) / max(tot, 1):5.1f} %   per round {int(t[i]) / max(rounds, 1):8.1f}   per pick {int(t[i]) / max(picks, 1):7.1f}")
