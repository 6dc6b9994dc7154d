"""Dev tool (experiment): the LDS-weight conv kernel over the flat step table (k_conv_lw, csrc/spconv_lw.hip) against
the size-based default of gf_conv_fwd on the submanifold tables of S150k levels 1-2: back-to-back launches over
rotating buffers, result against a float64 gather-matmul on the device."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import geoformer_amd
geoformer_amd.configure_runtime()
from geoformer_amd import sparse, scene

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
levels = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,2".split(","))]
batch = scene.make_batch([scene.make_scene(150_000, 1234)])
coords = batch["voxel_locs"].int().cuda().contiguous()
shape = tuple(int(s) for s in batch["spatial_shape"])
CH = [16, 32, 48, 64, 80, 96, 112]


def timeit(fn, n=reps, warm=4):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def ref64(x, W, nbr, M, scale=None, shift=None, res=None, osc=None, osh=None):
    a = x.double()
    if scale is not None: a = torch.clamp_min(a * scale.double() + shift.double(), 0)
    out = torch.zeros(M, W.shape[2], dtype=torch.float64, device=x.device)
    for k in range(W.shape[0]):
        i = nbr[k, :M].long()
        ok = i >= 0
        out[ok] += a[i[ok]] @ W[k].double()
    if res is not None: out += res.double()
    if osc is not None: out = torch.clamp_min(out * osc.double() + osh.double(), 0)
    return out


sparse.FLAT_MIN_ROWS = 0
lv_coords, lv_shape = coords, shape
for lv in range(1, max(levels) + 1):
    M = lv_coords.shape[0]
    if lv in levels:
        C = CH[lv - 1]
        rules = sparse.subm_rules(lv_coords, sparse.build_index(lv_coords, 1, lv_shape))
        nbr = rules.nbr
        steps = int(rules.flat[0])
        print(f"== level {lv}: M={M} groups={(M + 15) // 16} steps={steps} C={C}", flush=True)
        NB = 4
        cfgs = ((C, C, False, True, False), (C, C, True, False, True), (2 * C, C, True, False, True))
        if os.environ.get("LW_CFG"): cfgs = tuple(cfgs[int(i)] for i in os.environ["LW_CFG"].split(","))
        for (cin, cout, aff, resid, oact) in cfgs:
            if cout > 32: continue
            xs = [torch.randn(M, cin, device="cuda") for _ in range(NB)]
            W = torch.randn(27, cin, cout, device="cuda") / np.sqrt(9 * cin)
            rs = [torch.randn(M, cout, device="cuda") for _ in range(NB)]
            outs = [torch.empty(M, cout, device="cuda") for _ in range(NB)]
            sc = torch.rand(cin, device="cuda") + 0.5; sh = torch.randn(cin, device="cuda") * 0.3
            osc = torch.rand(cout, device="cuda") + 0.5; osh = torch.randn(cout, device="cuda") * 0.3
            kw = {}
            if aff: kw.update(in_scale=sc, in_shift=sh)
            if oact: kw.update(out_scale=osc, out_shift=osh)
            r64 = ref64(xs[0], W, nbr, M, sc if aff else None, sh if aff else None, rs[0] if resid else None,
                        osc if oact else None, osh if oact else None)
            mflop = 2.0 * steps * 16 * cin * cout
            for name, flat in (("default", None), ("lw", rules.flat)):
                sparse.dev_conv_knobs(lw=1 if flat is not None else 0)
                def run(i):
                    k = dict(kw)
                    if resid: k["residual"] = rs[i % NB]
                    return sparse.conv_fwd(xs[i % NB], W, nbr, rules.gmask, 27, M, rules.ld, out=outs[i % NB], steps=rules.steps, flat=flat, **k)
                o = run(0).clone()
                err = float((o.double() - r64).abs().max())
                us = timeit(run)
                print(f"  {cin:3d}->{cout:3d} aff={int(aff)} res={int(resid)} oact={int(oact)}  {name:8s} {us:7.2f} us  padded-MFMA {mflop / us / 1e6 / 157.3:5.2f} of peak  err {err:.1e}", flush=True)
            sparse.dev_conv_knobs()
    if lv < max(levels):
        d = sparse.down_rules(lv_coords, 1, lv_shape)
        lv_coords, lv_shape = d.out_coords.contiguous(), d.out_shape
