"""GPU: the native U-Net executor (gf_unet_fwd, csrc/unet_exec.hip) against the per-module route it replaces
(input_conv -> UBlock tree -> output_layer through geoformer_amd.spconv, model/geoformer/geoformer.py:42-53,398-401):
both issue the same kernels, so the voxel features agree to rounding of the one fused output activation."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model():
    from geoformer_amd.model import GeoFormer, load_config
    from tests.util import synthetic_state_dict

    m = GeoFormer(load_config("test_geoformer_scannet.yaml"))
    m.load_state_dict(synthetic_state_dict(m.state_dict(), 3))
    m.cuda()
    m.eval()
    return m


def _backbone(m, batch, native):
    os.environ["GF_UNET_EXEC"] = "1" if native else "0"
    try:
        with torch.no_grad():
            feats, scores, _ = m.forward_backbone(batch, len(batch["offsets"]) - 1, want_preds=True)
        torch.cuda.synchronize()
        return feats, scores
    finally:
        os.environ.pop("GF_UNET_EXEC", None)


@pytest.mark.parametrize("scenes", [[("small", 8192, 5)], [("small", 6000, 7), ("small", 9000, 8)], [("room", 40000, 21)],
                                    [("room", 62000, 31), ("room", 58000, 32)]],
                         ids=["s8k", "batch2", "room40k", "batch2-level1-kernels"])
def test_unet_exec_matches_module_route(hip, scenes):
    from geoformer_amd import scene, unet_exec

    m = _model()
    mk = {"small": scene.make_small_scene, "room": scene.make_scene}
    batch = scene.make_batch([mk[k](n, s) for k, n, s in scenes])
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    x = m.preprocess_input(batch, len(scenes))
    assert unet_exec.supported(m, x.features, x.spatial_shape) is False  # autograd on: module route
    with torch.no_grad():
        assert unet_exec.supported(m, x.features, x.spatial_shape)
    f_ref, s_ref = _backbone(m, batch, native=False)
    f_nat, s_nat = _backbone(m, batch, native=True)
    assert f_nat.shape == f_ref.shape
    assert float((f_nat - f_ref).abs().max()) <= 2e-6 * max(1.0, float(f_ref.abs().max()))
    assert float((s_nat - s_ref).abs().max()) <= 1e-5 * max(1.0, float(s_ref.abs().max()))
    # a second call re-uses the cached parameter structs; an in-place parameter update rebuilds them
    f_again, _ = _backbone(m, batch, native=True)
    assert torch.equal(f_again, f_nat)
    with torch.no_grad():
        m.unet.blocks.block0.conv_branch[2].weight.mul_(1.5)
    f_ref2, _ = _backbone(m, batch, native=False)
    f_nat2, _ = _backbone(m, batch, native=True)
    assert float((f_nat2 - f_ref2).abs().max()) <= 2e-6 * max(1.0, float(f_ref2.abs().max()))
    assert float((f_nat2 - f_nat).abs().max()) > 1e-3


def test_unet_exec_training_mode_bn_takes_module_route(hip):
    from geoformer_amd import scene, unet_exec

    m = _model()
    batch = scene.make_batch([scene.make_small_scene(4096, 2)])
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    x = m.preprocess_input(batch, 1)
    m.unet.blocks.block0.conv_branch[0].train()
    with torch.no_grad():
        assert not unet_exec.supported(m, x.features, x.spatial_shape)


def _train_backbone(m, batch, native, seed):
    """One training forward + backward of the backbone (input_conv -> unet -> output_layer -> a fixed random
    projection as the loss) through the native training route or the module tree; returns output features, every
    parameter's gradient and the BatchNorm running statistics afterwards."""
    os.environ["GF_UNET_TRAIN_EXEC"] = "1" if native else "0"
    try:
        m.zero_grad(set_to_none=True)
        x = m.preprocess_input(batch, len(batch["offsets"]) - 1)
        x = m.unet_features(x, len(batch["offsets"]) - 1)
        g = torch.Generator(device="cuda").manual_seed(seed)
        w = torch.randn(x.features.shape, device="cuda", generator=g)
        (x.features * w).sum().backward()
        torch.cuda.synchronize()
        mods = (m.input_conv, m.unet, m.output_layer)
        grads = {n: p.grad.clone() for mod in mods for n, p in mod.named_parameters()}
        stats = {n: b.clone() for mod in mods for n, b in mod.named_buffers() if b.dtype == torch.float32}
        return x.features.detach().clone(), grads, stats
    finally:
        os.environ.pop("GF_UNET_TRAIN_EXEC", None)


@pytest.mark.parametrize("scenes", [[("small", 8192, 5)], [("small", 6000, 7), ("small", 9000, 8)],
                                    [("room", 62000, 31), ("room", 58000, 32)]], ids=["s8k", "batch2", "batch2-rooms"])
def test_unet_train_exec_matches_module_route(hip, scenes):
    """gf_unet_train_fwd / _bwd (csrc/unet_train.hip) against the module tree in training mode: the same kernels in the
    same order, so features and BatchNorm statistics agree to the last bits and the gradients to the summation order of
    the weight gradient's atomics."""
    import copy

    from geoformer_amd import scene, unet_train

    from geoformer_amd.model import GeoFormer, load_config
    from tests.util import synthetic_state_dict

    m = GeoFormer(load_config("geoformer_scannet.yaml", batch_size=len(scenes)))  # the training yaml: nothing frozen
    m.load_state_dict(synthetic_state_dict(m.state_dict(), 3))
    m.cuda()
    m.train()
    for mod in m.modules():  # (the voxel transformers' dropout would make two forwards differ)
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    mk = {"small": scene.make_small_scene, "room": scene.make_scene}
    batch = scene.make_batch([mk[k](n, s) for k, n, s in scenes])
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    assert unet_train.supported(m, m.preprocess_input(batch, len(scenes)))
    state = copy.deepcopy(m.state_dict())
    f_ref, g_ref, s_ref = _train_backbone(m, batch, native=False, seed=1)
    m.load_state_dict(state)  # the running statistics back to where they were
    f_nat, g_nat, s_nat = _train_backbone(m, batch, native=True, seed=1)
    assert f_nat.shape == f_ref.shape
    assert float((f_nat - f_ref).abs().max()) <= 2e-6 * max(1.0, float(f_ref.abs().max()))
    for n in s_ref:
        assert float((s_nat[n] - s_ref[n]).abs().max()) <= 1e-6 * max(1.0, float(s_ref[n].abs().max())), n
    assert set(g_nat) == set(g_ref)
    top = max(float(g.norm()) for g in g_ref.values())
    for n, gr in g_ref.items():
        gn = g_nat[n]
        assert gn.shape == gr.shape, n
        ref = float(gr.norm())
        # (the 1x1x1 identity convolutions are library GEMMs on the module route and the MFMA kernel here: rounding-level
        # differences in the features, amplified where a 10^5-row sum cancels
        # -- BatchNorm's gamma / beta gradients: 1.2e-3 .. 3.0e-3 at 120k rows, depending on the summation order inside
        # the deep levels' convolution kernels, which both routes share)
        assert float((gn - gr).norm()) <= 5e-3 * max(ref, 1e-5 * top), (n, float((gn - gr).norm()), ref)
    # eval-mode BatchNorm or a frozen parameter: not this route
    m.unet.blocks.block0.conv_branch[0].eval()
    assert not unet_train.supported(m, m.preprocess_input(batch, len(scenes)))


def test_unet_train_exec_sees_reallocated_tensors_and_replaced_modules(hip):
    """The layer program bakes addresses in: after a first native training forward, (i) tensors that are
    re-allocated -- a BatchNorm's running statistics and gamma through ``.data =``, a convolution weight through
    load_state_dict(assign=True) -- must be the ones the next forward reads and updates, and (ii) a BatchNorm replaced by
    a SyncBatchNorm1d must take the model off the native route."""
    import copy

    from geoformer_amd import parallel, scene, unet_train
    from geoformer_amd.model import GeoFormer, load_config
    from tests.util import synthetic_state_dict

    m = GeoFormer(load_config("geoformer_scannet.yaml", batch_size=1))
    m.load_state_dict(synthetic_state_dict(m.state_dict(), 3))
    m.cuda()
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    batch = scene.make_batch([scene.make_small_scene(8192, 5)])
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    state = copy.deepcopy(m.state_dict())
    _train_backbone(m, batch, native=True, seed=1)  # compiles the program
    prog0 = m.__dict__["_gf_unet_train_prog"]
    m.load_state_dict(state)
    # (i) new storage for some tensors, new VALUES in them: the old addresses keep the old values (the program's
    # references keep them allocated), so a forward that still read them would reproduce the first run
    bn = m.unet.blocks.block0.conv_branch[0]
    with torch.no_grad():
        bn.weight.data = bn.weight.data.clone() * 1.5
        bn.running_mean.data = bn.running_mean.data.clone() + 0.25
        bn.running_var.data = bn.running_var.data.clone() * 2.0
    w = m.unet.blocks.block1.conv_branch[2].weight
    sd = {k: v for k, v in m.state_dict().items()}
    sd["unet.blocks.block1.conv_branch.2.weight"] = (w.detach() * 0.5).clone()
    m.load_state_dict(sd, assign=True)
    for p in m.parameters():
        p.requires_grad_(True)
    state2 = copy.deepcopy(m.state_dict())
    f_ref, g_ref, s_ref = _train_backbone(m, batch, native=False, seed=1)
    m.load_state_dict(state2)
    assert unet_train.supported(m, m.preprocess_input(batch, 1))
    f_nat, g_nat, s_nat = _train_backbone(m, batch, native=True, seed=1)
    assert m.__dict__["_gf_unet_train_prog"] is not prog0
    assert float((f_nat - f_ref).abs().max()) <= 2e-6 * max(1.0, float(f_ref.abs().max()))
    for n in s_ref:
        assert float((s_nat[n] - s_ref[n]).abs().max()) <= 1e-6 * max(1.0, float(s_ref[n].abs().max())), n
    for n in ("blocks.block0.conv_branch.0.weight", "blocks.block1.conv_branch.2.weight"):  # (names inside m.unet)
        gr, gn = g_ref[n], g_nat[n]
        assert float((gn - gr).norm()) <= 5e-3 * max(float(gr.norm()), 1e-8), n
    # (ii) SyncBatchNorm conversion AFTER a first native forward: the structural answer is not a stale cache entry
    parallel.convert_sync_batchnorm(m)
    assert not unet_train.supported(m, m.preprocess_input(batch, 1))
