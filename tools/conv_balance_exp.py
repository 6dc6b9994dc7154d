"""Dev tool: does the order of the row groups matter?  Times the S150k level-1..3 submanifold convs on row-permuted
copies of the same problem (natural order, groups sorted by active-offset count, shuffled)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import sparse, scene


def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def permuted(nbr, gmask, M, ld, unit, order):
    """order: permutation of the full units (M // unit of them); the ragged tail keeps its place."""
    nfull = M // unit
    rows = (order[:, None] * unit + torch.arange(unit, device=order.device)[None, :]).reshape(-1)
    p = torch.cat([rows, torch.arange(nfull * unit, M, device=order.device)])  # new row -> old row
    inv = torch.empty_like(p); inv[p] = torch.arange(M, device=p.device)
    nb = nbr[:, :M][:, p].long()
    nb2 = torch.where(nb >= 0, inv[nb.clamp(min=0)], nb).int()
    out = torch.full_like(nbr, -1); out[:, :M] = nb2
    g = gmask.clone()
    gpu_ = unit // 16
    gv = g[: nfull * gpu_].view(nfull, gpu_)
    g[: nfull * gpu_] = gv[order].reshape(-1)
    return p, out.contiguous(), g.contiguous()


sc = scene.make_scene(150_000, 1234)
batch = scene.make_batch([sc])
coords = batch["voxel_locs"].int().cuda().contiguous()
shape = tuple(int(s) for s in batch["spatial_shape"])
levels = []
c, s = coords, shape
r = sparse.subm_rules(c, sparse.build_index(c, 1, s)); levels.append((r, c.shape[0]))
for L in range(3):
    d = sparse.down_rules(c, 1, s)
    c, s = d.out_coords.contiguous(), d.out_shape
    levels.append((sparse.subm_rules(c, d.index_out), d.M_out))
for L, (r, M) in enumerate(levels):
    C = 16 * (L + 1)
    unit = 32 if L == 0 else 16
    x = torch.randn(M, C, device="cuda"); W = torch.randn(27, C, C, device="cuda") * 0.05
    gm = r.gmask[: (M + 15) // 16].cpu().numpy().view(np.uint32)
    pop = np.array([bin(int(v)).count("1") for v in gm])
    nfull = M // unit
    if unit == 32:
        g2 = gm[: nfull * 2].reshape(nfull, 2)
        wt = np.array([bin(int(a | b)).count("1") for a, b in g2])
    else:
        wt = pop[:nfull]
    print(f"level {L+1}: M {M} C {C} units {nfull} weight mean {wt.mean():.1f} std {wt.std():.1f} max {wt.max()} "
          f"p90 {np.percentile(wt, 90):.0f}")
    y0 = sparse.conv_fwd(x, W, r.nbr, r.gmask, 27, M, r.ld)
    wt_t = torch.from_numpy(wt.astype(np.int64)).cuda()
    orders = {
        "natural": torch.arange(nfull, device="cuda"),
        "sorted_desc": torch.argsort(wt_t, descending=True, stable=True),
        "sorted_asc": torch.argsort(wt_t, stable=True),
        "shuffled": torch.randperm(nfull, device="cuda"),
    }
    # coarse sort: keep natural order inside 4 weight classes (locality inside a class)
    cls = torch.bucketize(wt_t.float(), torch.quantile(wt_t.float(), torch.tensor([0.25, 0.5, 0.75], device="cuda")))
    orders["classes4_desc"] = torch.argsort(-cls, stable=True)
    for name, order in orders.items():
        p, nb, g = permuted(r.nbr, r.gmask, M, r.ld, unit, order)
        xp = x[p].contiguous()
        y = sparse.conv_fwd(xp, W, nb, g, 27, M, r.ld)
        err = float((y - y0[p]).abs().max())
        us = timeit(lambda: sparse.conv_fwd(xp, W, nb, g, 27, M, r.ld))
        print(f"   {name:14s} {us:6.2f} us   (check {err:.1e})")
