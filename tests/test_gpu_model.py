"""GPU: the build's GeoFormer (HIP operators) against fixtures produced by the REFERENCE's own
forward() on the same scene, weights and numpy RNG state (tests/golden/make_golden.py).

Integer outputs (foreground set, FPS indices, reach sets) are compared bit-exactly; floats to
1e-4 abs (BASELINE.json north_star), a few accumulated quantities relative to their magnitude.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _to_dev(batch):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}


@pytest.fixture(scope="module", params=["1", "2", "0"], ids=["split-sampling", "two-stream", "sequential"])
def run(hip, request):
    os.environ["GF_OVERLAP"] = request.param  # sampling cut after the query picks with the BFS beside the rest / BFS beside grouping only / one after the other
    from geoformer_amd import scene
    from geoformer_amd.model import GeoFormer, load_config
    from tests.util import synthetic_state_dict

    z = np.load(os.path.join(G, "geoformer_s8k_eval.npz"))
    cfg = load_config("test_geoformer_scannet.yaml")
    torch.manual_seed(0)
    m = GeoFormer(cfg)
    m.load_state_dict(synthetic_state_dict(m.state_dict(), int(z["weight_seed"])))
    m.cuda()
    m.eval()
    batch = _to_dev(scene.make_batch([scene.make_small_scene(int(z["scene_points"]), int(z["scene_seed"]))]))
    cap = {}
    dec = m.forward_decoder

    def dec_w(cl, cf, ql, pc, geo, pei):
        cap["context_locs"], cap["context_feats"], cap["pre_enc_inds"] = cl.detach(), cf.detach(), pei.detach()
        cap["geo"] = geo[0]
        r = dec(cl, cf, ql, pc, geo, pei)
        cap["dec_outputs"] = r.detach()
        return r

    m.forward_decoder = dec_w
    np.random.seed(int(z["numpy_seed"]))
    with torch.no_grad():
        out = m(batch, 300, training=False)
    torch.cuda.synchronize()
    os.environ.pop("GF_OVERLAP", None)
    return z, out, cap, m


def test_backbone_semantic_scores(run):
    z, out, cap, m = run
    got = out["semantic_scores"].cpu().numpy()
    assert np.abs(got - z["semantic_scores"]).max() < 1e-4
    assert (out["fg_idxs"].cpu().numpy() == z["fg_idxs"]).all()  # identical foreground set


def test_aggregator_indices_and_features(run):
    z, out, cap, m = run
    assert (m.last_sampling_indices.cpu().numpy() == z["sampling_indices"]).all()  # same host RNG draw
    assert (cap["pre_enc_inds"].cpu().numpy() == z["pre_enc_inds"]).all()  # FPS bit-exact
    assert np.abs(cap["context_locs"].cpu().numpy() - z["context_locs"]).max() == 0
    assert np.abs(cap["context_feats"].cpu().numpy() - z["context_feats"]).max() < 1e-4


def test_geodesic_distances(run):
    z, out, cap, m = run
    geo = cap["geo"].cpu().numpy()
    assert ((geo >= 0).sum(1) == z["geo_reached"]).all()  # reach sets
    assert (geo[::8, ::4] == z["geo_sub"]).all()  # bit-exact fp32 path sums
    assert np.abs(np.where(geo >= 0, geo, 0).astype(np.float64).sum(1) - z["geo_rowsum"]).max() < 1e-6


def test_decoder_and_mask_head(run):
    z, out, cap, m = run
    assert np.abs(cap["dec_outputs"].cpu().numpy() - z["dec_outputs"]).max() < 1e-4
    mp = out["mask_predictions"][-1]
    assert np.abs(mp["cls_logits"].cpu().numpy() - z["cls_logits"]).max() < 1e-4
    ml = mp["mask_logits"][0].cpu().numpy()
    assert np.abs(ml[::8, ::4] - z["mask_logits_sub"]).max() < 1e-4
    n = ml.shape[1]
    assert np.abs(ml.astype(np.float64).sum(1) - z["mask_logits_rowsum"]).max() < 1e-4 * n


def test_proposals(run):
    z, out, cap, m = run
    cls_final, scores_final, masks_final = out["proposal_scores"]
    assert (cls_final.cpu().numpy() == z["proposal_cls"]).all()
    assert np.abs(scores_final.cpu().numpy() - z["proposal_scores"]).max() < 1e-4
    # mask sizes may differ by the few points whose logit lies within the 1e-4 tolerance of the 0.5 cut
    d = np.abs(masks_final.sum(1).cpu().numpy() - z["proposal_npoints"])
    assert d.max() <= 3 and (d > 0).mean() < 0.1


def test_deferred_proposals_on_a_second_stream(run):
    """forward(..., defer_proposals=True) under a non-default stream: get() gives the proposals of the plain forward."""
    from geoformer_amd import scene
    from geoformer_amd.model.geoformer import PendingProposals

    z, out, cap, m = run
    batch = _to_dev(scene.make_batch([scene.make_small_scene(int(z["scene_points"]), int(z["scene_seed"]))]))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    np.random.seed(int(z["numpy_seed"]))
    with torch.no_grad(), torch.cuda.stream(s):
        out2 = m(batch, 300, training=False, defer_proposals=True)
    assert isinstance(out2["proposal_scores"], PendingProposals)
    cls2, sc2, pr2 = out2["proposal_scores"].get()
    torch.cuda.synchronize()
    cls1, sc1, pr1 = out["proposal_scores"]
    assert torch.equal(cls1, cls2) and torch.equal(pr1, pr2) and (sc1 - sc2).abs().max().item() < 1e-6


def test_forwards_are_bit_reproducible_across_streams_and_instances(hip):
    """The forward runs on three HIP streams (sampling / BFS / small side work): the same seeded forward must give
    identical bits every time, and a fresh model's FIRST forward (which derives all cached parameter copies) must
    already agree with its second."""
    from geoformer_amd import scene
    from geoformer_amd.model import GeoFormer, load_config
    from tests.util import synthetic_state_dict

    cfg = load_config("test_geoformer_scannet.yaml")
    batch = _to_dev(scene.make_batch([scene.make_small_scene(24000, 5)]))

    def outputs(m):
        np.random.seed(3)
        with torch.no_grad():
            out = m(batch, 300, training=False)
        mp = out["mask_predictions"][-1]
        return [out["semantic_scores"].clone(), out["fg_idxs"].clone(), mp["cls_logits"].clone(),
                mp["mask_logits"][0].clone()]

    ref = None
    for inst in range(3):
        m = GeoFormer(cfg)
        m.load_state_dict(synthetic_state_dict(m.state_dict(), 0))
        m.cuda().eval()
        for rep in range(4):
            got = outputs(m)
            if ref is None:
                ref = got
                assert not torch.isnan(ref[3]).any() and ref[1].numel() > 2048
            for a, b in zip(ref, got):
                assert a.shape == b.shape and torch.equal(a, b), (inst, rep)


def test_output_schema(run):
    z, out, cap, m = run
    assert set(out) == {"semantic_scores", "fg_idxs", "num_insts", "batch_idxs", "mask_predictions",
                        "proposal_scores"}  # geoformer.py:402-528
    assert out["num_insts"] == 256


def test_geoformer_fs_episode_gpu(hip):
    """Few-shot episode (BASELINE config 4 shape, 1-shot) through the HIP operators vs the reference golden."""
    from tests.util import check_fs_episode, run_fs_episode

    check_fs_episode(*run_fs_episode("cuda"))


def test_fused_caches_follow_parameter_updates(hip):
    """The fused inference paths keep derived copies of parameters (folded BatchNorm, packed weights, MLP chains):
    an in-place update (version bump) must be picked up by the next forward, a `.data` edit after
    invalidate_fused_caches()."""
    from geoformer_amd import scene
    from geoformer_amd.model import GeoFormer, load_config
    from tests.util import synthetic_state_dict

    m = GeoFormer(load_config("test_geoformer_scannet.yaml"))
    m.load_state_dict(synthetic_state_dict(m.state_dict(), 0))
    m.cuda().eval()
    batch = _to_dev(scene.make_batch([scene.make_small_scene(3000, 5)]))
    with torch.no_grad():
        s0 = m(batch, 0, training=False)["semantic_scores"].clone()
        m.semantic_linear.bias[2] += 3.0
        s1 = m(batch, 0, training=False)["semantic_scores"].clone()
        assert torch.allclose(s1[:, 2] - s0[:, 2], torch.full_like(s0[:, 2], 3.0), atol=1e-4)
        conv = m.unet.blocks.block0.conv_branch[2]
        conv.weight.mul_(0.5)
        s2 = m(batch, 0, training=False)["semantic_scores"].clone()
        assert (s2 - s1).abs().max() > 1e-3
        m.semantic_linear.bias.data[2] -= 3.0
        m.invalidate_fused_caches()
        s3 = m(batch, 0, training=False)["semantic_scores"]
        assert torch.allclose(s3[:, 2] - s2[:, 2], torch.full_like(s0[:, 2], -3.0), atol=1e-4)
