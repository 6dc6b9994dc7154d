"""Dev tool: the fused mask head alone at S150k sizes (nq=256, N=60k), a few launches, for rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoformer_amd import pointops
N, nq = 60108, 256
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
feat, coords, qxyz = r(N, 16), r(N, 3), r(nq, 3)
geo = torch.rand(nq, N, device="cuda", generator=g); geo[geo < 0.3] = -1
mx = torch.rand(nq, device="cuda", generator=g)
w1, b1, w2, b2 = r(nq, 16, 19), r(nq, 16), r(nq, 16), r(nq)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    o = pointops.mask_head(feat, coords, geo, qxyz, mx, w1, b1, w2, b2)
torch.cuda.synchronize(); print("done", float(o.abs().mean()))
