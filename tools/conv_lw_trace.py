"""Dev tool: per-wave cycle stamps of k_conv_lw (library built with -DLW_TRACE: tools/build_variant.sh)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import geoformer_amd
geoformer_amd.configure_runtime()
from geoformer_amd import sparse, scene, _lib

level = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nbins = 0
CH = [16, 32, 48, 64]
batch = scene.make_batch([scene.make_scene(150_000, 1234)])
c = batch["voxel_locs"].int().cuda().contiguous(); shp = tuple(int(s) for s in batch["spatial_shape"])
for lv in range(1, level):
    d = sparse.down_rules(c, 1, shp); c, shp = d.out_coords.contiguous(), d.out_shape
M = c.shape[0]; C = CH[level - 1]
sparse.FLAT_MIN_ROWS = 0
rules = sparse.subm_rules(c, sparse.build_index(c, 1, shp))
flat = sparse.flat_steps(rules.nbr, rules.gmask, 27, M, rules.ld, nbins)
x = torch.randn(M, C, device="cuda"); W = torch.randn(27, C, C, device="cuda") * 0.05; res = torch.randn(M, C, device="cuda")
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
sparse.dev_conv_knobs(lw=1)
for i in range(3): sparse.conv_fwd(x, W, rules.nbr, rules.gmask, 27, M, rules.ld, residual=res, flat=flat)
torch.cuda.synchronize()
raw.gf_dev_lw_trace(ctypes.c_void_p(buf.data_ptr()))
sparse.conv_fwd(x, W, rules.nbr, rules.gmask, 27, M, rules.ld, residual=res, flat=flat)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 8); t = t[t[:, 0] != 0]
q = lambda a: "min %d  p10 %d  p50 %d  p90 %d  max %d" % (a.min(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max())
print(f"level {level}: waves {len(t)}")
print("to the barrier   ", q(t[:, 1] - t[:, 0])); print("barrier wait     ", q(t[:, 2] - t[:, 1])); print("loop             ", q(t[:, 3] - t[:, 2]))
print("whole wave       ", q(t[:, 3] - t[:, 0]))
print("steps per wave   ", q(t[:, 4])); print("groups per wave  ", q(t[:, 5]))
print("loop cycles per step", q((t[:, 3] - t[:, 2]) / np.maximum(t[:, 4], 1)))
