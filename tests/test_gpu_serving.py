"""GPU: the staggered serving loop (geoformer_amd/serving.py) and the three-part forward under it.

``GeoFormer.forward_split`` issues the launches of ``forward`` (the cross-attention in its 8-wave workgroup shape) and
``StaggeredForward`` only re-orders them across scenes, so every scene must come out as from a plain forward: integers
(foreground set, proposal classes and memberships) bit-exact, floats to the rounding of the 8-wave soft-max merge."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _to_dev(batch):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}


def _model():
    from geoformer_amd.model import GeoFormer, load_config
    from tests.util import synthetic_state_dict

    m = GeoFormer(load_config("test_geoformer_scannet.yaml"))
    m.load_state_dict(synthetic_state_dict(m.state_dict(), 0))
    m.cuda()
    m.eval()  # (returns None, like the reference's train())
    return m


def _key(out):
    mp = out["mask_predictions"][-1]
    cls, sc, pr = out["proposal_scores"] if isinstance(out["proposal_scores"], tuple) else out["proposal_scores"].get()
    return (out["fg_idxs"].clone(), mp["cls_logits"].clone(), mp["mask_logits"][0].clone(),
            cls if isinstance(cls, list) else cls.clone(), sc if isinstance(sc, list) else sc.clone(),
            pr if isinstance(pr, list) else pr.clone())


def _same(a, b):
    assert torch.equal(a[0], b[0])  # foreground set
    scale = max(1.0, float(a[2].abs().max()))
    assert float((a[1] - b[1]).abs().max()) <= 1e-5 * max(1.0, float(a[1].abs().max()))
    assert float((a[2] - b[2]).abs().max()) <= 1e-5 * scale
    if isinstance(a[3], list):
        assert isinstance(b[3], list) and a[3] == b[3]
        return
    assert torch.equal(a[3], b[3])
    assert float((a[4] - b[4]).abs().max()) <= 1e-5
    # memberships may differ on the few points whose logit sits within the float tolerance of the 0.5 cut
    d = (a[5] != b[5]).sum(1)
    assert int(d.max()) <= 3


def test_cross_attn_8_wave_shape_matches_16(hip):
    """gf_decoder_cross_attn_cfg with 8 waves per workgroup against the 16-wave launch on random operands."""
    from geoformer_amd import pointops

    g = torch.Generator(device="cuda").manual_seed(1)
    B, nq, nc, d = 2, 37, 500, 64
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
    geo = r(B, nq, nc).abs()
    geo[:, :, ::7] = -1.0  # unreachable contexts take the positional fall-back
    args = (geo, r(B, nq).abs(), r(B, nq, 3), r(B, nc, 3), r(B, 3) - 2, r(B, 3) + 2, r(3, 32), r(B, nq, d), r(B, nc, d),
            r(B, nc, d))
    wpack = pointops.decoder_pack_weights(r(d, d) * 0.1, r(d, d) * 0.1, r(d, d) * 0.1)
    b2 = r(d)
    ref = pointops.decoder_cross_attn(*args, wpack, b2)
    with pointops.co_resident_launches():
        got = pointops.decoder_cross_attn(*args, wpack, b2)
    again = pointops.decoder_cross_attn(*args, wpack, b2)
    assert torch.equal(ref, again)  # the context restores the 16-wave shape
    assert float((got - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))


def test_forward_split_equals_forward_on_the_golden_scene(hip):
    from geoformer_amd import scene

    z = np.load(os.path.join(G, "geoformer_s8k_eval.npz"))
    m = _model()
    batch = _to_dev(scene.make_batch([scene.make_small_scene(int(z["scene_points"]), int(z["scene_seed"]))]))
    np.random.seed(7)
    with torch.no_grad():
        ref = _key(m(batch, 300, training=False))
    lane = torch.cuda.Stream()
    lane.wait_stream(torch.cuda.current_stream())
    np.random.seed(7)
    with torch.no_grad(), torch.cuda.stream(lane):
        h = m.forward_split(batch, 300, training=False, defer_proposals=True)
        assert h.backbone_done is not None and h.outputs is None
        h.advance()
        assert len(h.stretch_done) >= 2 and h.outputs is None  # end of the sampling + the scene's BFS
        out = h.finish()
        assert h.finish() is out  # idempotent
    got = _key(out)
    torch.cuda.synchronize()
    _same(ref, got)
    # a forward that ends early: the handle has its outputs after the first part, the later calls are no-ops
    with torch.no_grad():
        h = m.forward_split(batch, 0, training=False)  # epoch <= prepare_epochs: backbone + semantic head only
    assert h.outputs is not None and "semantic_scores" in h.outputs and h.backbone_done is None
    assert h.advance().stretch_done == () and h.finish() is h.outputs


@pytest.mark.parametrize("phased", [True, False], ids=["phased-backbone", "whole-backbone"])
def test_staggered_loop_equals_one_scene_at_a_time(hip, phased):
    """Seven scenes of different sizes through StaggeredForward (each scene's decoder under the next scene's sampling /
    BFS stretch; phased: the next scene's first U-Net level under this scene's read-back and first picks, the hand-over
    called from inside gf_unet_fwd_phased) against plain forwards, in submission order, twice through one loop object."""
    from geoformer_amd import scene, serving

    m = _model()
    sizes = [(9000, 3), (24000, 5), (8192, 11), (15000, 2), (30000, 9), (12000, 4), (20000, 8)]
    batches = [_to_dev(scene.make_batch([scene.make_small_scene(n, s)])) for n, s in sizes]
    refs = []
    for i, b in enumerate(batches):
        np.random.seed(100 + i)
        with torch.no_grad():
            refs.append(_key(m(b, 300, training=False)))
    torch.cuda.synchronize()
    loop = serving.StaggeredForward(m, "cuda", phased=phased)
    for rep in range(2):
        outs = []
        for i, b in enumerate(batches):
            outs += loop.submit(b, seed=100 + i)
        outs += loop.drain()
        torch.cuda.synchronize()
        assert len(outs) == len(batches)
        for r, o in zip(refs, outs):
            _same(r, _key(o))


def test_staggered_loop_with_scenes_that_end_early(hip):
    """A scene without predicted foreground leaves the forward behind the backbone (mask_predictions None): the loop
    must hand it back in order, and the scenes around it must come out as from plain forwards."""
    from geoformer_amd import scene, serving

    m = _model()
    full = _to_dev(scene.make_batch([scene.make_small_scene(12000, 4)]))
    other = _to_dev(scene.make_batch([scene.make_small_scene(9000, 3)]))
    # a scene whose every point is predicted background: the backbone's class scores of a marked batch are pushed to the
    # first (non-object) classes before the foreground selection reads them
    empty = dict(other)
    empty["_all_background"] = True
    backbone = m.forward_backbone

    def marked(batch_input, *a, **k):
        res = backbone(batch_input, *a, **k)
        if batch_input.get("_all_background"):
            res[1][:, :4] += 1e4
        return res

    m.forward_backbone = marked
    with torch.no_grad():
        probe = m(empty, 300, training=False)
    assert probe["mask_predictions"] is None
    np.random.seed(1)
    with torch.no_grad():
        ref_full = _key(m(full, 300, training=False))
    np.random.seed(2)
    with torch.no_grad():
        ref_other = _key(m(other, 300, training=False))
    torch.cuda.synchronize()
    loop = serving.StaggeredForward(m, "cuda")
    order = [("full", full, 1), ("empty", empty, 5), ("other", other, 2), ("full", full, 1), ("empty", empty, 6)]
    outs = []
    for _, b, seed in order:
        outs += loop.submit(b, seed=seed)
    outs += loop.drain()
    torch.cuda.synchronize()
    assert len(outs) == len(order)
    for (name, _, _), o in zip(order, outs):
        if name == "empty":
            assert o["mask_predictions"] is None and "semantic_scores" in o
        else:
            _same(ref_full if name == "full" else ref_other, _key(o))


def test_staggered_loop_over_the_module_route(hip, monkeypatch):
    """GF_UNET_EXEC=0: the backbone runs as the module tree, nothing consumes the loop's phase request -- the hand-over
    then happens right behind the scene's own backbone, and every scene still comes out as from a plain forward."""
    from geoformer_amd import scene, serving

    monkeypatch.setenv("GF_UNET_EXEC", "0")
    m = _model()
    batches = [_to_dev(scene.make_batch([scene.make_small_scene(n, s)])) for n, s in [(9000, 3), (12000, 4), (8192, 11)]]
    refs = []
    for i, b in enumerate(batches):
        np.random.seed(50 + i)
        with torch.no_grad():
            refs.append(_key(m(b, 300, training=False)))
    torch.cuda.synchronize()
    loop = serving.StaggeredForward(m, "cuda")
    outs = []
    for i, b in enumerate(batches):
        outs += loop.submit(b, seed=50 + i)
    outs += loop.drain()
    torch.cuda.synchronize()
    assert len(outs) == len(batches)
    for r, o in zip(refs, outs):
        _same(r, _key(o))
