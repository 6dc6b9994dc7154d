import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoformer_amd import sparse, scene, _lib
lib = _lib.load()
lib.gf_dev_conv_occupancy.restype = ctypes.c_int
for b in (64, 128, 256, 512): print("occupancy API blocks/CU at block", b, lib.gf_dev_conv_occupancy(b))
sc = scene.make_scene(150_000, 1234)
batch = scene.make_batch([sc])
coords = batch["voxel_locs"].int().cuda().contiguous()
shape = tuple(int(s) for s in batch["spatial_shape"])
M = coords.shape[0]
rules = sparse.subm_rules(coords, sparse.build_index(coords, 1, shape))
x = torch.randn(M, 16, device="cuda"); W = torch.randn(27, 16, 16, device="cuda") * 0.05
def timeit(fn, n=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for b in (64, 128, 256, 512):
    os.environ["GF_CONV_BLOCK"] = str(b)
    print("block", b, round(timeit(lambda: sparse.conv_fwd(x, W, rules.nbr, rules.gmask, 27, M, rules.ld)), 1), "us")
os.environ["GF_CONV_BLOCK"] = "256"
for sp in ("0", "1"):
    os.environ["GF_CONV_SPLIT"] = sp
    print("split", sp, round(timeit(lambda: sparse.conv_fwd(x, W, rules.nbr, rules.gmask, 27, M, rules.ld)), 1), "us")
import numpy as np
gm = rules.gmask.cpu().numpy().view(np.uint32)
pop = np.array([bin(int(v)).count("1") for v in gm])
print("popcount hist", np.bincount(pop, minlength=28).tolist())
