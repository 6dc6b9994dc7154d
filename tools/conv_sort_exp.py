"""Dev tool (experiment): does ordering the level-1 rows by their 27-bit neighbour mask pay in the conv kernel?
The rows of the S150k scene are PHYSICALLY permuted (coords + features) by a stable sort on a key derived from the mask,
inside windows of W rows; the unchanged rulebook builder and conv kernel then see 16-row groups of similar rows.
Prints steps per group and the kernel time (events over a batch of launches, and per-launch from rocprofv3 if run under it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import sparse, scene

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
batch = scene.make_batch([scene.make_scene(150_000, 1234)])
coords0 = batch["voxel_locs"].int().contiguous()
shape = tuple(int(s) for s in batch["spatial_shape"])
M = coords0.shape[0]
r0 = sparse.subm_rules(coords0.cuda(), sparse.build_index(coords0.cuda(), 1, shape))
nbr = r0.nbr[:, :M].cpu().numpy()
bits = ((nbr >= 0) * (1 << np.arange(27, dtype=np.int64))[:, None]).sum(0)
R = int((nbr >= 0).sum())

def key_bits(sel):
    k = np.zeros(M, np.int64)
    for j, b in enumerate(sel):
        k |= ((bits >> b) & 1) << j
    return k
face = [4, 22, 10, 16, 12, 14]
edges = [1, 3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25]
def plane_key():
    k = np.zeros(M, np.int64); j = 0
    for axis in range(3):
        for v in range(3):
            sel = [kk for kk in range(27) if [kk // 9, (kk // 3) % 3, kk % 3][axis] == v and kk != 13]
            a = np.zeros(M, bool)
            for b in sel: a |= ((bits >> b) & 1).astype(bool)
            k |= a.astype(np.int64) << j; j += 1
    return k
keys = {"none": None, "face6": key_bits(face), "plane9": plane_key(), "face+edge18": key_bits(face + edges), "full27": bits}

def timeit(fn, n=reps, warm=5):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

NB = 6
W = torch.randn(27, 16, 16, device="cuda") * 0.05
res = [torch.randn(M, 16, device="cuda") for _ in range(NB)]
outs = [torch.empty(M, 16, device="cuda") for _ in range(NB)]
xs = [torch.randn(M, 16, device="cuda") for _ in range(NB)]
byt = 4 * (R * 16 + M * 16 + 27 * 256) + 8 * R + 4 * M * 16
for kname, key in keys.items():
    for win in ((M,) if key is None else (1024, 4096, 16384, M)):
        if key is None:
            order = np.arange(M)
        else:
            order = np.concatenate([s + np.argsort(key[s:s + win], kind="stable") for s in range(0, M, win)])
        c = coords0[torch.from_numpy(order)].contiguous().cuda()
        rules = sparse.subm_rules(c, sparse.build_index(c, 1, shape))
        gm = rules.gmask[: (M + 15) // 16].cpu().numpy().view(np.uint32)
        pc = np.array([bin(int(x)).count("1") for x in gm])
        def run(i):
            return sparse.conv_fwd(xs[i % NB], W, rules.nbr, rules.gmask, 27, M, rules.ld, out=outs[i % NB], steps=rules.steps,
                                   residual=res[i % NB])
        us = timeit(run)
        print(f"{kname:12s} window {win:7d}: steps/group {pc.mean():5.2f}  >12: {(pc > 12).mean():.3f}  {us:6.2f} us  "
              f"{byt / us / 1e6:5.2f} TB/s  frac {byt / us / 1e6 / 8:.3f}", flush=True)
