"""Dev tool: tiny sparse convs (one 16-row group) with different step counts, for rocprofv3 --kernel-trace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoformer_amd import sparse
M = 16
coords = torch.tensor([[0, i % 3, (i // 3) % 3, i // 9] for i in range(M)], dtype=torch.int32).cuda()
ix = sparse.build_index(coords, 1, (4, 4, 4))
r = sparse.subm_rules(coords, ix)
for C in (16, 48, 112):
    x = torch.randn(M, C, device="cuda"); W = torch.randn(27, C, C, device="cuda") * 0.05; W1 = torch.randn(1, C, C, device="cuda")
    for _ in range(6):
        y = sparse.conv_fwd(x, W, r.nbr, r.gmask, 27, M, r.ld)
    torch.cuda.synchronize()
    for _ in range(6):
        y = sparse.conv_fwd(x, W1, None, None, 1, M, 0)
    torch.cuda.synchronize()
print("done")
