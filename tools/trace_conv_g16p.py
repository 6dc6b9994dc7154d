"""Dev tool: per-wave cycle stamps of the pipelined level-1 conv (library built with -DCONV_TRACE)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import sparse, scene, _lib
lib = _lib.load()
batch = scene.make_batch([scene.make_scene(150_000, 1234)])
coords = batch["voxel_locs"].int().cuda().contiguous()
shape = tuple(int(s) for s in batch["spatial_shape"])
M = coords.shape[0]
rules = sparse.subm_rules(coords, sparse.build_index(coords, 1, shape))
x = torch.randn(M, 16, device="cuda"); W = torch.randn(27, 16, 16, device="cuda") * 0.05
out = torch.empty(M, 16, device="cuda")
tr = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.gf_dev_conv_trace.argtypes = [ctypes.c_void_p]
sparse.dev_conv_knobs(g16=1, g16_ldsw=1, g16_pipe=1)
for i in range(5):
    sparse.conv_fwd(x, W, rules.nbr, rules.gmask, 27, M, rules.ld, out=out, steps=rules.steps)
torch.cuda.synchronize()
assert raw.gf_dev_conv_trace(ctypes.c_void_p(tr.data_ptr())) == 0
sparse.conv_fwd(x, W, rules.nbr, rules.gmask, 27, M, rules.ld, out=out, steps=rules.steps)
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(-1, 8)[:2048]
t = t[t[:, 7] > 0]
t0 = t[:, 0].min()
f = 100.0  # s_memtime ticks at 100 MHz: 10 ns per tick
print("waves", t.shape[0], "groups/wave", t[:, 7].mean())
for name, v in (("start", t[:, 0] - t0), ("after barrier", t[:, 1] - t0), ("first data", t[:, 2] - t0), ("end", t[:, 6] - t0)):
    print(f"{name:14s} min {v.min()/f:7.2f} med {np.median(v)/f:7.2f} max {v.max()/f:7.2f} us")
for name, col in (("wait first row", 3), ("mfma section", 4), ("store", 5)):
    per = t[:, col] / t[:, 7]
    print(f"{name:14s} per group: med {np.median(per)/f:6.3f} mean {per.mean()/f:6.3f} max {per.max()/f:6.3f} us; per wave total med {np.median(t[:, col])/f:6.2f} us")
life = (t[:, 6] - t[:, 0]) / f
print("wave lifetime med", np.median(life), "max", life.max())
# per-wave composition of its chunk vs lifetime
st = rules.steps.cpu().numpy()
ng = (M + 15) // 16
tail = st[(rules.ld // 16) * 7 * 64:]
nch = int(tail[0]); bounds = tail[1:nch + 2]
gm = rules.gmask.cpu().numpy().view(np.uint32)[:ng]
pop = np.array([bin(int(x)).count("1") for x in gm])
tt = tr.cpu().numpy().reshape(-1, 8)[:nch]
life = (tt[:, 6] - tt[:, 0]).astype(np.float64)
steps_w = np.array([pop[bounds[c]:bounds[c + 1]].sum() for c in range(nch)])
heavy_w = np.array([(pop[bounds[c]:bounds[c + 1]] > 12).sum() for c in range(nch)])
cnt_w = np.array([bounds[c + 1] - bounds[c] for c in range(nch)])
ok = tt[:, 7] > 0
print("lifetime percentiles (cycles)", np.percentile(life[ok], [1, 10, 50, 90, 99, 100]).round())
print("steps/wave percentiles", np.percentile(steps_w[ok], [1, 50, 99, 100]), "groups/wave", np.percentile(cnt_w[ok], [1, 50, 99, 100]), "heavy/wave", np.percentile(heavy_w[ok], [50, 99, 100]))
A = np.stack([np.ones(ok.sum()), steps_w[ok], cnt_w[ok], heavy_w[ok]], 1)
coef, *_ = np.linalg.lstsq(A, life[ok], rcond=None)
print("lifetime ~ %.0f + %.1f*steps + %.1f*groups + %.1f*heavy ; resid std %.0f" % (*coef, (life[ok] - A @ coef).std()))
slow = np.argsort(-life * ok)[:8]
for c in slow: print("  slow chunk", c, "life", life[c], "groups", cnt_w[c], "steps", steps_w[c], "heavy", heavy_w[c], "wait", tt[c, 3], "mfma", tt[c, 4], "store", tt[c, 5], "t_barrier", tt[c,1]-tt[c,0], "t_first", tt[c,2]-tt[c,0])
