"""Dev tool: would hipGraph replay of the training step's two launch-bound transformer stacks pay?  The deep U-Net levels'
voxel transformer (4 scenes padded to L tokens) and the 4-layer decoder (B 4, nq 256, nc 2048), forward + backward, eager
against torch.cuda.make_graphed_callables: host time until the calls return and time until the device is done."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import geoformer_amd
geoformer_amd.configure_runtime()
import numpy as np, torch, torch.nn as nn
from geoformer_amd.model import GeoFormer, load_config
from geoformer_amd.model.layers import RelPosSpec
from tests.util import synthetic_state_dict

dev = torch.device("cuda", 0)
cfg = load_config("geoformer_scannet.yaml", batch_size=4, prepare_epochs=120)
m = GeoFormer(cfg); m.load_state_dict(synthetic_state_dict(m.state_dict(), 0)); m.to(dev); m.train()


class Layers(nn.Module):
    def __init__(self, tr):
        super().__init__(); self.tr = tr
    def forward(self, x, mask):
        for layer in self.tr.layers: x = layer(x, mask=mask)
        return self.tr.norm(x)


class Dec(nn.Module):
    def __init__(self, dec, gauss_B):
        super().__init__(); self.dec = dec; self.gauss_B = gauss_B
    def forward(self, tgt, memory, qpos, geo, max_geo, ql, cl, lo, hi):
        rp = RelPosSpec(geo, max_geo, ql, cl, lo, hi, self.gauss_B)
        return self.dec(tgt=tgt, memory=memory, pos=None, query_pos=qpos, relative_pos=rp)


def bench(name, fn, args, n=20):
    def it():
        out = fn(*args)
        out.square().mean().backward()
    for _ in range(3): it()
    torch.cuda.synchronize(); th = tt = 0.0
    for _ in range(n):
        t0 = time.perf_counter(); it(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        th += t1 - t0; tt += t2 - t0
    print(f"{name:34s} host {th / n * 1e3:6.2f} ms   done {tt / n * 1e3:6.2f} ms", flush=True)


which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("bt", "both"):
    tr = m.unet
    while tr.transformer is None: tr = tr.u
    for L in (64, 160):
        lay = Layers(tr.transformer)
        x = torch.randn(4, L, 128, device=dev, requires_grad=True)
        mask = torch.ones(4, 1, L, device=dev, dtype=torch.long); mask[1, 0, L // 2:] = 0
        bench(f"voxel transformer L={L} eager", lay, (x, mask))
        g = torch.cuda.make_graphed_callables(lay, (x.detach().clone().requires_grad_(True), mask.clone()), allow_unused_input=True)
        bench(f"voxel transformer L={L} graphed", g, (x, mask))
        # same numbers?  (dropout makes them differ: compare in eval-mode dropout p=0 by zeroing p)
if which in ("dec", "both"):
    B, nq, nc, d = 4, 256, 2048, 64
    dec = Dec(m.decoder, m.pos_embedding.gauss_B.contiguous())
    mem = torch.randn(nc, B, d, device=dev, requires_grad=True)
    tgt = torch.randn(nq, B, d, device=dev, requires_grad=True)
    qpos = torch.randn(nq, B, d, device=dev, requires_grad=True)
    geo = torch.rand(B, nq, nc, device=dev); geo[geo > 0.8] = -1
    mg = geo.max(dim=2)[0].contiguous()
    cl = torch.rand(B, nc, 3, device=dev) * 4; ql = cl[:, :nq].contiguous()
    lo = torch.zeros(B, 3, device=dev); hi = torch.full((B, 3), 4.0, device=dev)
    args = (tgt, mem, qpos, geo, mg, ql, cl, lo, hi)
    bench("decoder eager", dec, args)
    sample = tuple(a.detach().clone().requires_grad_(a.requires_grad) for a in args)
    g = torch.cuda.make_graphed_callables(dec, sample, allow_unused_input=True)
    bench("decoder graphed", g, args)
