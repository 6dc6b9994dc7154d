"""Training form of the decoder's token-side stages (gf_decoder_pre_train_* / gf_decoder_post_train_*,
csrc/decoder_layer_train.hip) against the arithmetic of TransformerDecoderLayer.forward_pre_rel
(model/transformer_detr.py:425-463) in float64 with the kernels' own dropout masks, and the whole training decoder
against the framework-module route."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _layer(seed, ff=256):
    from geoformer_amd.model.layers import TransformerDecoderLayer

    torch.manual_seed(seed)
    layer = TransformerDecoderLayer(64, nhead=4, dim_feedforward=ff, dropout=0.1, use_rel=True)
    norm = nn.LayerNorm(64)
    with torch.no_grad():
        for p in list(layer.parameters()) + list(norm.parameters()):
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
            else:
                p.uniform_(-0.3, 0.3)
        for m in [layer.norm1, layer.norm2, layer.norm3, norm]:
            m.weight.uniform_(0.5, 1.5)
    return layer, norm


def _close(a, b, what, tol=2e-4):
    b = b.detach().to(torch.float64)
    scale = max(1.0, float(b.abs().max()))
    err = float((a.detach().double() - b).abs().max())
    assert err <= tol * scale, f"{what}: max |diff| {err:.3g} at scale {scale:.3g}"


def _ln(x, m):
    return F.layer_norm(x, (64,), m.weight.double(), m.bias.double(), m.eps)


@pytest.mark.parametrize("B,T,p", [(2, 256, 0.0), (3, 100, 0.1), (1, 17, 0.1)])
def test_decoder_pre_stage_matches_float64(B, T, p):
    from geoformer_amd import pointops

    dev = torch.device("cuda", 0)
    layer, norm = _layer(1)
    layer, norm = layer.to(dev), norm.to(dev)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, T, 64, generator=g).to(dev).requires_grad_(True)
    qpos = torch.randn(B, T, 64, generator=g).to(dev).requires_grad_(True)
    wa, wb = torch.randn(B, T, 64, generator=g).to(dev), torch.randn(B, T, 64, generator=g).to(dev)
    pre, _ = pointops.decoder_stage_tensors(layer, norm)
    seed, li = 777, 2
    t2n, q1 = pointops.decoder_pre_train(x, qpos, li, p, seed, pre)
    got = torch.autograd.grad((t2n * wa).sum() + (q1 * wb).sum(), [x, qpos] + pre)
    # float64 reference with the same masks
    D = lambda t: t.double()
    sa = layer.self_attn
    rows = (torch.arange(B, device=dev).view(B, 1) * T + torch.arange(T, device=dev).view(1, T))  # [B,T]
    ch = torch.arange(64, device=dev)
    t2 = _ln(D(x), layer.norm1)
    qk = t2 + D(qpos)
    Wi, bi = D(sa.in_proj_weight), D(sa.in_proj_bias)
    q = (qk @ Wi[:64].t() + bi[:64]).view(B, T, 4, 16).transpose(1, 2)
    k = (qk @ Wi[64:128].t() + bi[64:128]).view(B, T, 4, 16).transpose(1, 2)
    v = (t2 @ Wi[128:].t() + bi[128:]).view(B, T, 4, 16).transpose(1, 2)
    s = torch.softmax((q * 0.25) @ k.transpose(2, 3), dim=-1)  # [B,4,T,T]
    keep = pointops.dropout_keep_reference(seed, p, 8 * li + 0, rows.view(B, 1, T, 1).expand(B, 4, T, T),
                                           torch.arange(T, device=dev).view(1, 1, 1, T) * 4 + torch.arange(4, device=dev).view(1, 4, 1, 1))
    o = ((s * keep.double()) @ v).transpose(1, 2).reshape(B, T, 64)
    att = o @ D(sa.out_proj.weight).t() + D(sa.out_proj.bias)
    x1 = D(x) + att * pointops.dropout_keep_reference(seed, p, 8 * li + 1, rows.view(B, T, 1).expand(B, T, 64), ch.view(1, 1, 64)).double()
    r_t2n = _ln(x1, layer.norm2)
    r_q1 = r_t2n @ D(layer.attn_mlp[0].weight).t() + D(layer.attn_mlp[0].bias)
    want = torch.autograd.grad((r_t2n * wa.double()).sum() + (r_q1 * wb.double()).sum(), [x, qpos] + pre)
    _close(t2n, r_t2n, "t2n")
    _close(q1, r_q1, "q1")
    for i, (a, b) in enumerate(zip(got, want)):
        _close(a, b, f"gradient {i}")


@pytest.mark.parametrize("B,T,p,ff", [(2, 256, 0.0, 256), (3, 100, 0.1, 256), (1, 17, 0.1, 64)])
def test_decoder_post_stage_matches_float64(B, T, p, ff):
    from geoformer_amd import pointops

    dev = torch.device("cuda", 0)
    layer, norm = _layer(3, ff)
    layer, norm = layer.to(dev), norm.to(dev)
    g = torch.Generator().manual_seed(4)
    ca = torch.randn(B, T, 64, generator=g).to(dev).requires_grad_(True)
    t2n = torch.randn(B, T, 64, generator=g).to(dev).requires_grad_(True)
    wa, wb = torch.randn(B, T, 64, generator=g).to(dev), torch.randn(B, T, 64, generator=g).to(dev)
    _, post = pointops.decoder_stage_tensors(layer, norm)
    seed, li = 4242, 1
    x3, inter = pointops.decoder_post_train(ca, t2n, li, p, seed, post)
    got = torch.autograd.grad((x3 * wa).sum() + (inter * wb).sum(), [ca, t2n] + post)
    got_inter_only = torch.autograd.grad(
        (pointops.decoder_post_train(ca, t2n, li, p, seed, post)[1] * wb).sum(), [ca, t2n] + post)
    D = lambda t: t.double()
    rows = (torch.arange(B, device=dev).view(B, 1) * T + torch.arange(T, device=dev).view(1, T)).view(B, T, 1)
    keep = lambda site, n: pointops.dropout_keep_reference(seed, p, 8 * li + site, rows.expand(B, T, n),
                                                           torch.arange(n, device=dev).view(1, 1, n)).double()
    y = torch.relu(D(ca) @ D(layer.out_mlp[0].weight).t() + D(layer.out_mlp[0].bias))
    x2 = y + D(t2n) * keep(2, 64)
    t3 = _ln(x2, layer.norm3)
    h = torch.relu(t3 @ D(layer.linear1.weight).t() + D(layer.linear1.bias)) * keep(3, ff)
    r_x3 = x2 + (h @ D(layer.linear2.weight).t() + D(layer.linear2.bias)) * keep(4, 64)
    r_inter = _ln(r_x3, norm)
    want = torch.autograd.grad((r_x3 * wa.double()).sum() + (r_inter * wb.double()).sum(), [ca, t2n] + post, retain_graph=True)
    want_inter_only = torch.autograd.grad((r_inter * wb.double()).sum(), [ca, t2n] + post)
    _close(x3, r_x3, "x3")
    _close(inter, r_inter, "inter")
    for i, (a, b) in enumerate(zip(got, want)):
        _close(a, b, f"gradient {i}")
    for i, (a, b) in enumerate(zip(got_inter_only, want_inter_only)):
        _close(a, b, f"gradient {i} (x3 unused)")


def _decoder_case(dev, B=2, nq=256, nc=2048):
    from geoformer_amd.model.layers import TransformerDecoder, TransformerDecoderLayer, RelPosSpec

    torch.manual_seed(5)
    dec = TransformerDecoder(TransformerDecoderLayer(64, nhead=4, dim_feedforward=256, dropout=0.1, use_rel=True), 4,
                             return_intermediate=True).to(dev)
    g = torch.Generator().manual_seed(6)
    mem = torch.randn(nc, B, 64, generator=g).to(dev).requires_grad_(True)
    qpos = torch.randn(nq, B, 64, generator=g).to(dev).requires_grad_(True)
    geo = torch.rand(B, nq, nc, generator=g).to(dev)
    geo[geo > 0.8] = -1
    mg = geo.max(dim=2)[0].contiguous()
    cl = (torch.rand(B, nc, 3, generator=g) * 4).to(dev)
    ql = cl[:, :nq].contiguous()
    lo, hi = torch.zeros(B, 3, device=dev), torch.full((B, 3), 4.0, device=dev)
    gauss_B = torch.randn(3, 32, generator=g).to(dev).contiguous()
    rp = RelPosSpec(geo, mg, ql, cl, lo, hi, gauss_B)
    return dec, mem, qpos, rp


def test_training_decoder_matches_the_module_route_without_dropout():
    dev = torch.device("cuda", 0)
    dec, mem, qpos, rp = _decoder_case(dev)
    dec.eval()  # dropout off, gradients on
    g = torch.Generator().manual_seed(7)
    res = []
    w = None
    for flag in ("1", "0"):
        os.environ["GF_FUSED_DECODER_TRAIN"] = flag
        try:
            out = dec(tgt=mem[:256], memory=mem, query_pos=qpos, relative_pos=rp)
            if w is None:
                w = torch.randn(out.shape, generator=g).to(dev)
            params = list(dec.parameters())
            grads = torch.autograd.grad((out * w).sum(), [mem, qpos] + params, allow_unused=True)
            res.append((out.detach().clone(), grads))
        finally:
            os.environ.pop("GF_FUSED_DECODER_TRAIN", None)
    (oa, ga), (ob, gb) = res
    assert oa.shape == ob.shape
    _close(oa, ob, "decoder output", 3e-4)
    for i, (a, b) in enumerate(zip(ga, gb)):
        assert (a is None) == (b is None), i
        if a is not None:
            _close(a, b, f"gradient {i}", 1e-3)


def test_training_decoder_with_dropout_is_reproducible_under_the_framework_seed():
    dev = torch.device("cuda", 0)
    dec, mem, qpos, rp = _decoder_case(dev, B=1, nq=64, nc=256)
    dec.train()
    outs = []
    for s in (11, 11, 12):
        torch.manual_seed(s)
        out = dec(tgt=mem[:64], memory=mem, query_pos=qpos, relative_pos=rp)
        outs.append((out.detach().clone(), torch.autograd.grad(out.square().sum(), [mem])[0].clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    # (the cross-attention's key-side gradients are accumulated with atomics: same masks, last-bit differences)
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-6 * float(outs[0][1].abs().max()))
    assert not torch.equal(outs[0][0], outs[2][0])
