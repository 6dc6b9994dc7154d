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


def test_early_exit_branches_gpu(hip):
    """forward()'s early exits through the fused GPU paths (geoformer.py:423-439,451-452): backbone-only epochs, an
    EMPTY predicted-foreground set (the fused selection's count read-back is 0; eval, deferred and training mode)."""
    from geoformer_amd import scene
    from geoformer_amd.model import GeoFormer, load_config
    from tests.util import synthetic_state_dict

    m = GeoFormer(load_config("test_geoformer_scannet.yaml"))
    m.load_state_dict(synthetic_state_dict(m.state_dict(), 0))
    m.cuda().eval()
    batch = _to_dev(scene.make_batch([scene.make_small_scene(6000, 5)]))
    with torch.no_grad():
        early = m(batch, m.prepare_epochs, training=False)
        assert set(early) == {"semantic_scores"}
        m.semantic_linear.bias[:4] += 1e4  # nothing is predicted as an object class
        empty = m(batch, 300, training=False)
        assert empty["mask_predictions"] is None and "proposal_scores" not in empty
        empty_d = m(batch, 300, training=False, defer_proposals=True)
        assert empty_d["mask_predictions"] is None
    m.train()
    out = m(batch, 300, training=True)
    assert out["mask_predictions"] is None


def test_geoformer_fs_episode_gpu(hip):
    """Few-shot episode (BASELINE config 4 shape, 1-shot) through the HIP operators vs the reference golden."""
    from tests.util import check_fs_episode, run_fs_episode

    check_fs_episode(*run_fs_episode("cuda"))


def test_geoformer_fs_5shot_episode_gpu_matches_oracle_backend(hip, oracle):
    """BASELINE config 4 as named: 1-way 5-shot = process_support on five full support scenes, the mean embedding,
    one GeoFormerFS.forward(..., training=False, support_embeddings=mean) on the query scene (yaml copy with
    k_shot: 5; test_fs.py:157-174, geoformer_fs.py:424-455).  The GPU episode against the same episode through the
    oracle's operators on the host: support embeddings, mask / class logits, similarity scores."""
    from geoformer_amd import scene
    from geoformer_amd.model import GeoFormerFS, load_config
    from oracle import cpu_backend
    from tests.util import synthetic_state_dict

    def dicts():
        q = scene.make_batch([scene.make_small_scene(8192, 7)])
        sups = [scene.make_batch([scene.make_small_scene(5000 + 400 * i, 20 + i)]) for i in range(5)]
        for d in [q] + sups:
            d["batch_offsets"] = d["offsets"]
        for d in sups:
            d["support_masks"] = (d["instance_labels"] >= 0).long()
        return q, sups

    def episode(device):
        m = GeoFormerFS(load_config("test_geoformer_fs_scannet.yaml", k_shot=5))
        m.load_state_dict(synthetic_state_dict(m.state_dict(), 2))
        m.semantic_linear.bias.data[3] += 1.0
        m.to(device)
        m.eval()
        q, sups = dicts()
        mv = lambda d: {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in d.items()}  # noqa: E731
        cap = []
        orig = m.get_mask_prediction

        def gmp(*a, **k):
            r = orig(*a, **k)
            cap.append(r[-1]["mask_logits"][0].detach().cpu())
            return r

        m.get_mask_prediction = gmp
        np.random.seed(11)
        with torch.no_grad():
            embs = [m.process_support(mv(d), training=False) for d in sups]
            emb = torch.stack(embs).mean(0)
            out = m(None, mv(q), training=False, remember=False, support_embeddings=emb)
        scores, props = out["proposal_scores"]
        return (torch.stack(embs).cpu(), out["semantic_scores"].cpu(), m.cache_data[3].cpu(), cap[0],
                scores.cpu() if len(scores) else torch.zeros(0), props.sum(1).cpu() if len(scores) else torch.zeros(0))

    with cpu_backend.installed():
        ref = episode("cpu")
    got = episode("cuda")
    assert (got[0] - ref[0]).abs().max() < 1e-4  # the five support embeddings
    assert (got[1] - ref[1]).abs().max() < 1e-4
    assert torch.equal(got[2], ref[2])  # foreground set
    assert (got[3] - ref[3]).abs().max() < 2e-4 * max(1.0, float(ref[3].abs().max()))  # mask logits
    assert got[4].shape == ref[4].shape
    if len(ref[4]):
        assert (got[4] - ref[4]).abs().max() < 1e-4
        assert (got[5] - ref[5]).abs().max() <= 3


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


def test_two_host_threads_on_their_own_streams_share_one_model(hip):
    """The forward parks events and side results of its BFS / aux streams on the model between its stages; that state
    is keyed by the caller's stream, so two host threads running different scenes through ONE model on their own
    streams must each get exactly what a serial run gives."""
    import threading

    from geoformer_amd import scene
    from geoformer_amd.model import GeoFormer, load_config
    from tests.util import synthetic_state_dict

    m = GeoFormer(load_config("test_geoformer_scannet.yaml"))
    m.load_state_dict(synthetic_state_dict(m.state_dict(), 0))
    m.cuda().eval()
    batches = [_to_dev(scene.make_batch([scene.make_small_scene(n, s)])) for n, s in ((9000, 5), (7000, 6))]

    def run(b, seed):
        np.random.seed(seed)  # (the host draw is numpy's global generator: the threads below draw under a lock)
        with torch.no_grad():
            o = m(b, 300, training=False)
        mp = o["mask_predictions"][-1]
        return [o["fg_idxs"].clone(), mp["cls_logits"].clone(), mp["mask_logits"][0].clone()]

    serial = [run(b, 1) for b in batches]
    torch.cuda.synchronize()
    results, errors = [None, None], []
    draw_lock = threading.Lock()
    orig_choice = __import__("geoformer_amd").pointops.legacy_choice

    def locked_choice(n, k, out=None):  # same generator state for every draw: reseed under the lock
        with draw_lock:
            np.random.seed(1)
            return orig_choice(n, k, out=out)

    import geoformer_amd.pointops as po

    orig_draw = po.draw_sample

    def locked_draw(n, k, xyz_src, bufs):  # (the eval forward's route: draw + upload + gather in one native call)
        with draw_lock:
            np.random.seed(1)
            return orig_draw(n, k, xyz_src, bufs)

    po.legacy_choice = locked_choice
    po.draw_sample = locked_draw
    try:
        def worker(i):
            try:
                st = torch.cuda.Stream()
                for _ in range(4):
                    with torch.cuda.stream(st), torch.no_grad():
                        o = m(batches[i], 300, training=False)
                        mp = o["mask_predictions"][-1]
                        results[i] = [o["fg_idxs"].clone(), mp["cls_logits"].clone(), mp["mask_logits"][0].clone()]
                    st.synchronize()
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    finally:
        po.legacy_choice = orig_choice
        po.draw_sample = orig_draw
    assert not errors, errors
    for i in range(2):
        for a, b in zip(serial[i], results[i]):
            assert a.shape == b.shape and torch.equal(a, b), i


def test_knn_truncation_is_reported_with_the_last_read_back(hip):
    """More than 1024 in-radius neighbours (1300 copies of one point): gf_knn_radius sets its device flag, the forward
    carries it to the proposals' read-back and raises there instead of returning distances over truncated rows."""
    from geoformer_amd import _lib, scene
    from geoformer_amd.model import GeoFormer, load_config
    from tests.util import synthetic_state_dict

    m = GeoFormer(load_config("test_geoformer_scannet.yaml"))
    m.load_state_dict(synthetic_state_dict(m.state_dict(), 0))
    with torch.no_grad():
        m.semantic_linear.bias[:4] -= 1e4  # every point is foreground
    m.cuda().eval()
    batch = _to_dev(scene.make_batch([scene.make_small_scene(8192, 7)]))
    batch["locs_float"][100:1400] = batch["locs_float"][50]
    np.random.seed(0)
    with torch.no_grad(), pytest.raises(_lib.GeoFormerHipError, match="in-radius"):
        m(batch, 300, training=False)


def test_fs_requery_many_equals_sequential_requeries(hip):
    """GeoFormerFS.requery_many -- E re-queries of a cached scene as ONE decoder pass with the episode as the batch
    index, one host synchronisation -- against E sequential forward(..., remember=True) calls and
    against the reference's own outputs for the two embeddings the FS golden holds (test_fs.py:157-174 is the loop
    this replaces).  The batched pass takes the multi-scene route of the decoder (other launch shapes, the projections
    as batched GEMMs): scores to 1e-5, point memberships identical up to a few threshold ties."""
    from tests.util import fs_dicts, run_fs_episode

    z, m, emb, out, out2, cap = run_fs_episode("cuda")
    sup, q = fs_dicts()
    q = _to_dev(q)
    embs = torch.cat([emb, emb * 0.5, emb * 0.0 + 0.3, -emb, emb * 0.5])
    with torch.no_grad():
        seq = [m(None, q, training=False, remember=True, support_embeddings=embs[i:i + 1])["proposal_scores"]
               for i in range(embs.shape[0])]
        many = m.requery_many(q, embs)
        m.REQUERY_CHUNK = 2  # the same in chunks of two episodes (three decoder passes)
        many2 = m.requery_many(q, embs)
    torch.cuda.synchronize()
    assert len(many) == len(seq) == len(many2)
    for got in (many, many2):
        for a, b in zip(seq, got):
            assert len(a[0]) == len(b[0])
            if len(a[0]):
                assert (a[0] - b[0]).abs().max().item() < 1e-5
                assert int((a[1] != b[1]).sum()) <= 3
    assert any(len(a[0]) for a in seq)  # at least one embedding yields proposals
    # the reference's GeoFormerFS on the same scene: embedding e (fresh forward) and 0.5 e (remember=True)
    for idx, ks, kn in ((0, "proposal_scores", "proposal_npoints"), (1, "proposal_scores_half", "proposal_npoints_half"),
                        (4, "proposal_scores_half", "proposal_npoints_half")):
        scores, props = many[idx]
        assert len(scores) == len(z[ks])
        if len(scores):
            assert np.abs(scores.cpu().numpy() - z[ks]).max() < 1e-4
            assert np.abs(props.sum(1).cpu().numpy() - z[kn]).max() <= 3


def test_fs_requery_many_against_the_oracle_backed_sequential_loop(hip):
    """Row f4 against the ORACLE, not against the HIP path itself: the same few-shot model on the host over the oracle's
    C operators (oracle/cpu_backend.py) runs the reference's loop -- one forward(..., remember=True) per support embedding
    (test_fs.py:157-174) -- and GeoFormerFS.requery_many on the GPU (one decoder pass, gf_mask_head_episodes,
    gf_proposal_stats_fs over all episodes) must reproduce every episode: the same accepted queries, scores to 1e-4,
    point memberships up to threshold ties.  Six embeddings in chunks of 6 / 4 / 1 episodes per pass."""
    from oracle import cpu_backend
    from tests.util import fs_dicts, run_fs_episode

    with cpu_backend.installed():
        zc, mc, emb_c, _, _, _ = run_fs_episode("cpu")
        _, qc = fs_dicts()
        scale = [1.0, 0.5, 0.75, -1.0, 1.5, 0.25]
        embs_c = torch.cat([emb_c * f for f in scale])
        with torch.no_grad():
            ref = [mc(None, qc, training=False, remember=True, support_embeddings=embs_c[i:i + 1])["proposal_scores"]
                   for i in range(len(scale))]
    z, m, emb, out, out2, cap = run_fs_episode("cuda")
    assert (emb.cpu() - emb_c).abs().max().item() < 1e-4
    _, q = fs_dicts()
    q = _to_dev(q)
    embs = embs_c.cuda()  # the host run's embeddings: both sides start from identical numbers
    assert sum(len(r[0]) > 0 for r in ref) >= 2  # (episodes with and without accepted proposals)
    for chunk in (16, 4, 1):
        m.REQUERY_CHUNK = chunk
        with torch.no_grad():
            many = m.requery_many(q, embs)
        torch.cuda.synchronize()
        assert len(many) == len(ref)
        for e, (a, b) in enumerate(zip(ref, many)):
            assert len(a[0]) == len(b[0]), (chunk, e, len(a[0]), len(b[0]))
            if len(a[0]):
                assert (a[0] - b[0].cpu()).abs().max().item() < 1e-4, (chunk, e)
                assert a[1].shape == b[1].shape
                assert int((a[1] != b[1].cpu()).sum()) <= 3, (chunk, e)


@pytest.mark.gpu
def test_backbone_transformer_padded_pass_equals_per_scene_loop():
    """layers.BackboneTransformer with several scenes on the GPU (one padded pass with the layers' key mask) against the
    per-scene loop (the route CPU tensors take): outputs and gradients, scenes of different sizes and an empty one."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from geoformer_amd.model.layers import BackboneTransformer

    torch.manual_seed(0)
    d = 32
    tr = BackboneTransformer(d, 2, 4, 64).eval()  # eval: dropout off (its draws differ between the two layouts)
    sizes = [37, 0, 120, 5]
    bid = torch.cat([torch.full((n,), b, dtype=torch.int32) for b, n in enumerate(sizes)])
    xyz = torch.randint(0, 40, (bid.numel(), 3)).float()
    feats = torch.randn(bid.numel(), d, requires_grad=True)
    gout = torch.randn(bid.numel(), d)
    ref = tr(xyz, feats, bid, batch_size=len(sizes))
    ref.backward(gout)
    g_ref = [feats.grad.clone()] + [p.grad.clone() for p in tr.parameters()]
    trg = BackboneTransformer(d, 2, 4, 64).eval()
    trg.load_state_dict(tr.state_dict())
    trg.cuda()
    fg = feats.detach().cuda().requires_grad_(True)
    out = trg(xyz.cuda(), fg, bid.cuda(), batch_size=len(sizes))
    out.backward(gout.cuda())
    assert (out.cpu() - ref).abs().max().item() < 2e-5
    g_got = [fg.grad.cpu()] + [p.grad.cpu() for p in trg.parameters()]
    for a, b in zip(g_got, g_ref):
        assert (a - b).abs().max().item() < 2e-4 * max(1.0, b.abs().max().item())


def test_forward_edge_case_scene_matches_oracle_backend(hip, oracle):
    """SURVEY 8(d)'s edge-case variants through the WHOLE eval forward (the operator tests hold them one by one): a scene
    with exact duplicate points (kNN / FPS / ball-query distance ties, rows of zero-length edges in the BFS), points inside
    |p|^2 <= 1e-3 (the sampler's skip rule, sampling_gpu.cu:104), in a batch of two scenes of different sizes; the GPU
    forward against the same forward through the oracle's operators on the host, integers bit-exact."""
    from geoformer_amd import scene
    from geoformer_amd.model import GeoFormer, load_config
    from oracle import cpu_backend
    from tests.util import synthetic_state_dict

    def scenes():
        a, b = scene.make_small_scene(8192, 9), scene.make_small_scene(5000, 10)
        a["xyz"][100:160] = a["xyz"][7]            # 60 copies of one point
        a["xyz"][300:304] = [[0.01, 0.0, 0.01], [0.0, 0.02, 0.0], [0.0, 0.0, 0.0], [-0.02, 0.01, 0.0]]  # |p|^2 <= 1e-3
        b["xyz"][::50] = b["xyz"][1::50]           # a duplicate in every 50th position
        return [a, b]

    def run(device):
        m = GeoFormer(load_config("test_geoformer_scannet.yaml", n_decode_point=512, n_query_points=64))
        m.load_state_dict(synthetic_state_dict(m.state_dict(), 0))
        with torch.no_grad():
            m.semantic_linear.bias[4:] += 2.0  # most points foreground: the special points take part
        m.to(device)
        m.eval()
        batch = scene.make_batch(scenes())
        batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        cap = {}
        dec = m.forward_decoder

        def dec_w(cl, cf, ql, pc, geo, pei):
            cap["pei"], cap["cl"] = pei.detach().cpu(), cl.detach().cpu()
            cap["geo"] = [g.detach().cpu() for g in geo]
            return dec(cl, cf, ql, pc, geo, pei)

        m.forward_decoder = dec_w
        np.random.seed(21)
        with torch.no_grad():
            out = m(batch, 300, training=False)
        return out, cap

    got, cg = run("cuda")
    with cpu_backend.installed():
        ref, cc = run("cpu")
    assert (got["semantic_scores"].cpu() - ref["semantic_scores"]).abs().max() < 1e-4
    assert torch.equal(got["fg_idxs"].cpu(), ref["fg_idxs"]) and got["fg_idxs"].numel() > 8000
    assert torch.equal(cg["pei"], cc["pei"]) and torch.equal(cg["cl"], cc["cl"])  # FPS picks of both scenes
    for a, b in zip(cg["geo"], cc["geo"]):
        assert torch.equal(a, b)  # reach sets and fp32 path sums, bit for bit
    mg, mc = got["mask_predictions"][-1], ref["mask_predictions"][-1]
    assert (mg["cls_logits"].cpu() - mc["cls_logits"]).abs().max() < 1e-4
    for a, b in zip(mg["mask_logits"], mc["mask_logits"]):
        assert (a.cpu() - b).abs().max() < 1e-4 * max(1.0, float(b.abs().max()))


def test_rulebooks_ahead_of_the_callers_stream_change_nothing(hip):
    """batch["inputs_event"] (GeoFormer._inputs_ahead -> gf_unet_fwd_ahead): the backbone's rulebooks are built on the
    executor's side stream behind the batch's own events instead of behind what the caller's stream has queued.  Scenes
    of different sizes alternate on one stream with a long-running kernel queued in front of every forward (so the
    rulebooks of scene i+1 really are built while scene i's tail and that kernel are still running, into the same
    workspace): every output equals the plain route's bit for bit; an event list that is not empty is waited for."""
    from geoformer_amd import scene
    from geoformer_amd.model import GeoFormer, load_config
    from tests.util import synthetic_state_dict

    m = GeoFormer(load_config("test_geoformer_scannet.yaml"))
    m.load_state_dict(synthetic_state_dict(m.state_dict(), 0))
    m.cuda().eval()
    hosts = [scene.make_batch([scene.make_small_scene(n, s)]) for n, s in ((24000, 5), (9000, 6), (16000, 7))]
    plain = [_to_dev(h) for h in hosts]
    busy = torch.randn(4096, 4096, device="cuda")

    def run(b, seed):
        np.random.seed(seed)
        with torch.no_grad():
            o = m(b, 300, training=False)
        mp = o["mask_predictions"][-1]
        return [o["semantic_scores"].clone(), o["fg_idxs"].clone(), mp["cls_logits"].clone(), mp["mask_logits"][0].clone()]

    ref = [run(b, 10 + i) for i, b in enumerate(plain)]
    torch.cuda.synchronize()
    ahead = [dict(b, inputs_event=()) for b in plain]
    for rep in range(3):
        for i, b in enumerate(ahead):
            for _ in range(6):
                busy = busy @ busy * 1e-3  # ~ms of work in front of the forward on the caller's stream
            got = run(b, 10 + i)
            for a, g in zip(ref[i], got):
                assert a.shape == g.shape and torch.equal(a, g), (rep, i)
    # coordinates that ARE produced by pending work, on another stream: the batch names the event
    other = torch.cuda.Stream()
    for i, h in enumerate(hosts):
        with torch.cuda.stream(other):
            for _ in range(4):
                busy2 = busy @ busy * 1e-3
            locs = h["voxel_locs"].cuda(non_blocking=True) + (busy2[0, 0] * 0).long()  # (queued behind the matmuls)
            ev = torch.cuda.Event()
            ev.record(other)
        b = dict(plain[i], voxel_locs=locs, inputs_event=(ev,))
        locs.record_stream(torch.cuda.current_stream())
        got = run(b, 10 + i)
        for a, g in zip(ref[i], got):
            assert torch.equal(a, g), i
