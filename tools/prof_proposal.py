"""Dev tool: the fused proposal kernels alone at S150k sizes, for rocprofv3 --kernel-trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoformer_amd import pointops
N, nq, ncls, npts = 60108, 256, 20, 150269
g = torch.Generator(device="cuda").manual_seed(0)
logits = torch.randn(nq, N, device="cuda", generator=g) * 3 - 2
cls_logits = torch.randn(nq, ncls, device="cuda", generator=g)
sem = torch.softmax(torch.randn(N, ncls, device="cuda", generator=g), 1)
fg = torch.sort(torch.randperm(npts, device="cuda", generator=g)[:N])[0]
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    cp, n, sc, fin = pointops.proposal_stats(logits, cls_logits, sem, 0.5, 0.2, 100)
    sel = torch.nonzero(fin).view(-1).int()
    pr = pointops.proposal_scatter(logits, sel, fg, 0.5, npts)
torch.cuda.synchronize(); print("done", int(sel.numel()))
