"""Training form of the deep levels' voxel transformer (gf_backbone_transformer_train_fwd / _bwd, csrc/backbone_attn.hip)
against the framework modules it replaces (before_transformer_linear -> BackboneTransformer -> after_transformer_linear:
model/geoformer/geoformer_modules.py:64-68,120-127, model/transformer.py:62-188), outputs and every gradient, without
dropout and with the kernel's own dropout masks rebuilt on the host and fed to the modules."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _stack(c, seed, n_layers=2):
    from geoformer_amd.model.layers import BackboneTransformer

    torch.manual_seed(seed)
    before, tr, after = nn.Linear(c, 128), BackboneTransformer(d_model=128, N=n_layers, heads=4, d_ff=64), nn.Linear(128, c)
    with torch.no_grad():
        for mod in tr.modules():  # Norm parameters away from their (1, 0) initialisation
            if hasattr(mod, "alpha"):
                mod.alpha.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.3, 0.3)
    return before, tr, after


def _scenes(lens, c, seed, dev):
    g = torch.Generator().manual_seed(seed)
    coords = []
    for b, n in enumerate(lens):
        xyz = torch.randint(0, 12, (n, 3), generator=g, dtype=torch.int32)
        coords.append(torch.cat([torch.full((n, 1), b, dtype=torch.int32), xyz], 1))
    coords = torch.cat(coords).contiguous()
    feats = torch.randn(coords.shape[0], c, generator=g)
    return feats.to(dev), coords.to(dev)


def _reference(before, tr, after, feats, coords, n_scenes, seed, p):
    """The modules' arithmetic in float64 with the given dropout masks (dropout_keep_reference)."""
    from geoformer_amd import pointops

    dd = torch.float64
    P = lambda t: t.to(dd)
    out = torch.zeros(feats.shape[0], after.out_features, dtype=dd, device=feats.device)
    ids = coords[:, 0].long()
    x_in = feats.to(dd)
    for b in range(n_scenes):
        rows = torch.nonzero(ids == b).flatten()
        if rows.numel() == 0:
            continue
        T = rows.numel()
        pts = coords[rows, 1:].to(dd)
        rel = (pts.unsqueeze(1) - pts.unsqueeze(0)).mean(dim=1)
        x = x_in[rows] @ P(before.weight).t() + P(before.bias) + rel @ P(tr.position_linear.weight).t() + P(tr.position_linear.bias)
        ch = torch.arange(128, device=feats.device)
        for l, layer in enumerate(tr.layers):
            def norm(nm, x):
                mu = x.mean(-1, keepdim=True)
                return P(nm.alpha) * (x - mu) / (x.std(-1, keepdim=True) + nm.eps) + P(nm.bias)
            a = layer.attn_1
            x2 = norm(layer.norm_1, x)
            q = (x2 @ P(a.q_linear.weight).t() + P(a.q_linear.bias)).view(T, 4, 32).transpose(0, 1)
            k = (x2 @ P(a.k_linear.weight).t() + P(a.k_linear.bias)).view(T, 4, 32).transpose(0, 1)
            v = (x2 @ P(a.v_linear.weight).t() + P(a.v_linear.bias)).view(T, 4, 32).transpose(0, 1)
            s = torch.softmax(q @ k.transpose(1, 2) / np.sqrt(32.0), dim=-1)  # [4, T, T]
            heads = torch.arange(4, device=feats.device).view(4, 1, 1)
            keys = torch.arange(T, device=feats.device).view(1, 1, T)
            keep = pointops.dropout_keep_reference(seed, p, 4 * l + 0, rows.view(1, T, 1).expand(4, T, T), keys * 4 + heads)
            o = ((s * keep.to(dd)) @ v).transpose(0, 1).reshape(T, 128)
            att = o @ P(a.out.weight).t() + P(a.out.bias)
            x = x + att * pointops.dropout_keep_reference(seed, p, 4 * l + 1, rows.view(T, 1).expand(T, 128), ch.view(1, 128)).to(dd)
            x2 = norm(layer.norm_2, x)
            hid = torch.relu(x2 @ P(layer.ff.linear_1.weight).t() + P(layer.ff.linear_1.bias))
            hid = hid * pointops.dropout_keep_reference(seed, p, 4 * l + 2, rows.view(T, 1).expand(T, 64), ch[:64].view(1, 64)).to(dd)
            f = hid @ P(layer.ff.linear_2.weight).t() + P(layer.ff.linear_2.bias)
            x = x + f * pointops.dropout_keep_reference(seed, p, 4 * l + 3, rows.view(T, 1).expand(T, 128), ch.view(1, 128)).to(dd)
        mu = x.mean(-1, keepdim=True)
        y = P(tr.norm.alpha) * (x - mu) / (x.std(-1, keepdim=True) + tr.norm.eps) + P(tr.norm.bias)
        out[rows] = y @ P(after.weight).t() + P(after.bias)
    return out


@pytest.mark.parametrize("lens,c,p,n_layers", [((37, 160, 5, 64), 96, 0.0, 2), ((37, 160, 5, 64), 112, 0.1, 2),
                                               ((1, 300), 192, 0.1, 2), ((16, 0, 33), 96, 0.1, 1)])
def test_voxel_transformer_train_matches_modules_float64(lens, c, p, n_layers):
    from geoformer_amd import pointops

    dev = torch.device("cuda", 0)
    before, tr, after = _stack(c, 3, n_layers)
    before, tr, after = before.to(dev), tr.to(dev), after.to(dev)
    for m in tr.modules():
        if isinstance(m, nn.Dropout):
            m.p = p
    tr.train(p > 0)
    feats, coords = _scenes(lens, c, 5, dev)
    feats.requires_grad_(True)
    assert pointops.backbone_transformer_train_supported(feats, coords, tr)
    seed = 1234567
    out = pointops.backbone_transformer_train(feats, coords, len(lens), before, tr, after, seed=seed)
    gen = torch.Generator(device="cpu").manual_seed(9)
    w = torch.randn(out.shape, generator=gen).to(dev)
    params = pointops.backbone_transformer_tensors(before, tr, after)
    got = torch.autograd.grad((out * w).sum(), [feats] + params)
    ref = _reference(before, tr, after, feats, coords, len(lens), seed, p)
    want = torch.autograd.grad((ref * w.double()).sum(), [feats] + params)
    def close(a, b, what):
        b = b.to(torch.float64)
        scale = max(1.0, float(b.detach().abs().max()))
        err = float((a.double() - b).abs().max())
        assert err <= 2e-4 * scale, f"{what}: max |diff| {err:.3g} at scale {scale:.3g}"
    close(out, ref, "output")
    names = ["feats"] + [f"param{i}" for i in range(len(params))]
    for n, a, b in zip(names, got, want):
        close(a, b, "gradient of " + n)


def test_voxel_transformer_train_is_deterministic_and_dropout_rate_is_p():
    from geoformer_amd import pointops

    dev = torch.device("cuda", 0)
    before, tr, after = [m.to(dev) for m in _stack(96, 4)]
    tr.train()
    feats, coords = _scenes((120, 90, 200, 77), 96, 6, dev)
    feats.requires_grad_(True)
    params = pointops.backbone_transformer_tensors(before, tr, after)
    runs = []
    for _ in range(2):
        out = pointops.backbone_transformer_train(feats, coords, 4, before, tr, after, seed=42)
        runs.append([out.detach().clone()] + [g.clone() for g in torch.autograd.grad(out.square().sum(), [feats] + params)])
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    other = pointops.backbone_transformer_train(feats, coords, 4, before, tr, after, seed=43)
    assert not torch.equal(other, runs[0][0])
    rows = torch.arange(4096, device=dev).view(-1, 1).expand(4096, 128)
    cols = torch.arange(128, device=dev).view(1, -1)
    keep = pointops.dropout_keep_reference(42, 0.1, 5, rows, cols)
    frac = float((keep == 0).float().mean())
    assert abs(frac - 0.1) < 0.004, frac
    assert float(keep.max()) == pytest.approx(1 / 0.9, rel=1e-6)


def test_training_program_uses_the_native_voxel_transformer():
    """The batch-4 U-Net training forward/backward with the native stack against the same step with the framework
    modules (GF_FUSED_VOXEL_TRANSFORMER=0), dropout off: features and parameter gradients agree to float tolerance."""
    import os
    from geoformer_amd import scene
    from geoformer_amd.model import GeoFormer, load_config
    from tests.util import synthetic_state_dict

    dev = torch.device("cuda", 0)
    cfg = load_config("geoformer_scannet.yaml", batch_size=2, prepare_epochs=120)
    m = GeoFormer(cfg)
    m.load_state_dict(synthetic_state_dict(m.state_dict(), 0))
    m.to(dev).train()
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    batch = scene.make_batch([scene.make_scene(20_000, 1), scene.make_scene(15_000, 2)])
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    res = []
    for flag in ("1", "0"):
        os.environ["GF_FUSED_VOXEL_TRANSFORMER"] = flag
        try:
            m.zero_grad(set_to_none=True)
            np.random.seed(0)
            out = m.forward_backbone(batch, 2)
            sem = out[1]  # semantic scores [N, classes]
            sem.square().mean().backward()
            res.append((sem.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
        finally:
            os.environ.pop("GF_FUSED_VOXEL_TRANSFORMER", None)
    a, b = res
    assert torch.allclose(a[0], b[0], rtol=2e-4, atol=2e-4 * float(b[0].abs().max()))
    assert a[1].keys() == b[1].keys()
    for n in a[1]:
        scale = float(b[1][n].abs().max())  # (gradients that are zero in exact arithmetic -- a bias in front of a
        assert float((a[1][n] - b[1][n]).abs().max()) <= 5e-4 * scale + 1e-7, n  # BatchNorm -- are rounding noise)
