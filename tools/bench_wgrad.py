"""Dev tool: the weight gradient on the level-1 table of the batch-4 training batch (523k voxels), dense walk vs the
group-mask walk, per channel pair."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene, sparse
dev = torch.device("cuda", 0)
b = scene.make_batch([scene.make_scene(int(n), 50 + i) for i, n in enumerate((150_000, 120_000, 180_000, 100_000))])
c = b["voxel_locs"].to(dev).int().contiguous()
shape = tuple(int(x) for x in b["spatial_shape"])
rules = sparse.subm_rules(c, sparse.build_index(c, 4, shape))
M = c.shape[0]
print("voxels", M, "mean offsets per group", float(torch.tensor([bin(int(x)).count("1") for x in rules.gmask[:2000].cpu().tolist()]).float().mean()))
for Cin, Cout in ((16, 16), (32, 16), (32, 32), (64, 32)):
    x = torch.randn(M, Cin, device=dev); g = torch.randn(M, Cout, device=dev)
    res = []
    for gm in (None, rules.gmask):
        for _ in range(2): w = sparse.conv_wgrad(x, g, rules.nbr, 27, M, rules.ld, gmask=gm)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): w = sparse.conv_wgrad(x, g, rules.nbr, 27, M, rules.ld, gmask=gm)
        torch.cuda.synchronize(); res.append(((time.perf_counter() - t) / 10 * 1e6, w))
    print(f"{Cin}->{Cout}: dense {res[0][0]:.0f} us, masked {res[1][0]:.0f} us, max diff {(res[0][1] - res[1][1]).abs().max().item():.2e} of {res[0][1].abs().max().item():.1f}")
