"""Dev tool: fused mask head forward / backward at the training size (nq=128, N=30000) and the eval size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoformer_amd import pointops
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for nq, N in ((128, 30000), (256, 60000)):
    feat = torch.randn(N, 16, device="cuda", requires_grad=True)
    params = (torch.randn(nq, 337, device="cuda") * 0.3).requires_grad_()
    coords = torch.rand(N, 3, device="cuda") * 6 - 3
    qxyz = coords[:nq].clone()
    geo = torch.rand(nq, N, device="cuda") * 5
    geo[torch.rand(nq, N, device="cuda") < 0.3] = -1
    mx = torch.sqrt(geo.max(1)[0])
    gout = torch.randn(nq, N, device="cuda")
    out = pointops.mask_head_train(feat, params, coords, geo, qxyz, mx)
    print(f"nq {nq} N {N}: fwd {t(lambda: pointops.mask_head_packed(feat.detach(), coords, geo, qxyz, mx, params.detach())):.1f} us",
          f"bwd {t(lambda: torch.autograd.grad(pointops.mask_head_train(feat, params, coords, geo, qxyz, mx), (feat, params), gout)):.1f} us (fwd+bwd)")
