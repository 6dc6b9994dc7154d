"""Generates the golden fixtures by running the REFERENCE's own Python on CPU.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
The reference model code is imported, never copied; its native dependencies are replaced by the
CPU oracle through tests/golden/ref_shims.py.  Outputs (committed, small):
    geoformer_state_dict_keys.json   parameter/buffer names and shapes of GeoFormer
    geoformer_s8k_eval.npz           S8k scene, test yaml (nq=256, nc=2048), eval forward: stage outputs
    geoformer_train_small.npz        (`train`) the TRAINING branch (geoformer.py:468-493) + InstSetCriterion + backward on a
                                     2-scene small batch, train yaml with batch_size 2 / dec_dropout 0 / nq 32 / nc 512 and
                                     the mask-head subsample cut to 2000 points so the host RNG draw is exercised:
                                     subsample indices, per-layer logits, loss dict, per-parameter gradient norms + samples
    geoformer_train_mid.npz          (`train_mid`) the same on two room-sized scenes (90k + 70k points) with the yaml's own
                                     nq=128 / nc=2048 and the reference's own 30 000-point subsample (GPU test only)
    geodesic_vectorize.npz           cal_geodesic_vectorize on a 3k-point cloud (pins the BFS oracle)
    decoder_layer.npz                one TransformerDecoderLayer + fourier embedding on random inputs
    matrix_nms.npz                   util.utils_3d.matrix_non_max_suppression on overlapping random proposals (`nms`)
Weights are NOT stored: both sides call tests.util.synthetic_state_dict (deterministic by name).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
_WHICH = [a for a in sys.argv[1:] if a in ("geodesic", "decoder", "model", "fs", "nms", "train", "train_mid")]
_TRAIN = _WHICH in (["train"], ["train_mid"])
_YAML = ("config/test_geoformer_fs_scannet.yaml" if _WHICH == ["fs"] else
         "config/geoformer_scannet.yaml" if _TRAIN else "config/test_geoformer_scannet.yaml")
sys.argv = ["make_golden", "--config", os.path.join(REF, _YAML)]  # util/config.py parses argv at import: one yaml per process

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.golden import ref_shims  # noqa: E402

ref_shims.install(REF)
os.chdir(REF)  # util/config.py and friends use relative paths

if _WHICH == ["fs"]:
    from model.geoformer.geoformer_fs import GeoFormerFS  # noqa: E402  (reference class)
else:
    from model.geoformer.geoformer import GeoFormer  # noqa: E402  (reference class)
from model.geoformer import geodesic_utils  # noqa: E402
from model.transformer_detr import TransformerDecoderLayer  # noqa: E402
from model.pos_embedding import PositionEmbeddingCoordsSine  # noqa: E402

from geoformer_amd import scene  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.util import synthetic_state_dict  # noqa: E402


def sub(a, rs=8, cs=4):
    return np.ascontiguousarray(a[::rs, ::cs])


def golden_model():
    torch.manual_seed(0)
    m = GeoFormer()
    sd = m.state_dict()
    json.dump({k: list(v.shape) for k, v in sd.items()}, open(os.path.join(HERE, "geoformer_state_dict_keys.json"), "w"),
              indent=0)
    m.load_state_dict(synthetic_state_dict(sd, 0))
    m.eval()
    sc = scene.make_small_scene(8192, 7)
    batch = scene.make_batch([sc])
    cap = {}
    # capture intermediate stages by wrapping the reference's own methods
    orig_agg, orig_dec, orig_geo = m.forward_aggregator, m.forward_decoder, geodesic_utils.cal_geodesic_vectorize

    def agg(*a, **k):
        r = orig_agg(*a, **k)
        cap["context_locs"], cap["context_feats"], cap["pre_enc_inds"] = [t.detach().numpy().copy() for t in r]
        return r

    def dec(context_locs, context_feats, query_locs, pc_dims, geo_dists, pre_enc_inds):
        cap["geo"] = geo_dists[0].numpy().copy()
        r = orig_dec(context_locs, context_feats, query_locs, pc_dims, geo_dists, pre_enc_inds)
        cap["dec_outputs"] = r.detach().numpy().copy()
        return r

    m.forward_aggregator, m.forward_decoder = agg, dec
    np.random.seed(5)
    rs = np.random.get_state()
    with torch.no_grad():
        out = m(batch, 300, training=False)
    np.random.set_state(rs)
    n_fg = int(out["fg_idxs"].shape[0])
    sampling_indices = np.random.choice(n_fg, min(n_fg, 50000), replace=False)  # what forward_aggregator drew
    mp = out["mask_predictions"][-1]
    geo = cap["geo"]
    ml = mp["mask_logits"][0].numpy()
    cls_final, scores_final, masks_final = out["proposal_scores"]
    np.savez_compressed(
        os.path.join(HERE, "geoformer_s8k_eval.npz"),
        scene_seed=7, scene_points=8192, numpy_seed=5, weight_seed=0,
        semantic_scores=out["semantic_scores"].numpy(), fg_idxs=out["fg_idxs"].numpy(),
        sampling_indices=sampling_indices, pre_enc_inds=cap["pre_enc_inds"],
        context_locs=cap["context_locs"], context_feats=cap["context_feats"],
        geo_sub=sub(geo), geo_rowsum=np.where(geo >= 0, geo, 0).astype(np.float64).sum(1),
        geo_reached=(geo >= 0).sum(1), dec_outputs=cap["dec_outputs"],
        cls_logits=mp["cls_logits"].numpy(), mask_logits_sub=sub(ml),
        mask_logits_rowsum=ml.astype(np.float64).sum(1),
        proposal_cls=np.asarray(cls_final), proposal_scores=np.asarray(scores_final),
        proposal_npoints=np.asarray(masks_final).sum(1) if len(cls_final) else np.zeros(0),
    )
    print("model golden: N", batch["locs"].shape[0], "N_fg", n_fg, "proposals", len(cls_final))


def golden_geodesic():
    sc = scene.make_small_scene(3000, 21)  # ~4 cm spacing: radius 0.12 links ~25 neighbours
    rng = np.random.default_rng(3)
    pts = np.ascontiguousarray(sc["xyz"][rng.permutation(sc["xyz"].shape[0])[:3000]])
    pts[40:44] = pts[5]  # exact duplicates: ties and the "self is not rank 0" case
    nq = 16
    n = pts.shape[0]
    pre_enc = torch.from_numpy(np.concatenate([[5, 41], rng.permutation(n)[:62]]).astype(np.int64))[None]
    idx = ref_shims._FaissIndex(None, 3, None)
    geo = geodesic_utils.cal_geodesic_vectorize(idx, pre_enc, torch.from_numpy(pts), torch.tensor([0, n]),
                                                max_step=256, neighbor=64, radius=0.12, n_queries=nq)[0].numpy()
    geo5 = geodesic_utils.cal_geodesic_vectorize(idx, pre_enc, torch.from_numpy(pts), torch.tensor([0, n]),
                                                 max_step=5, neighbor=64, radius=0.12, n_queries=nq)[0].numpy()
    np.savez_compressed(os.path.join(HERE, "geodesic_vectorize.npz"), points=pts, pre_enc_inds=pre_enc.numpy(),
                        radius=0.12, neighbor=64, n_queries=nq, geo_max256=geo, geo_max5=geo5)
    print("geodesic golden: reached", (geo >= 0).mean(), "max", geo.max())


def golden_decoder_layer():
    torch.manual_seed(1)
    d, nq, nc, B = 64, 24, 96, 2
    layer = TransformerDecoderLayer(d_model=d, nhead=4, dim_feedforward=64, dropout=0.1, normalize_before=True,
                                    use_rel=True)
    sd = synthetic_state_dict(layer.state_dict(), 3)
    layer.load_state_dict(sd)
    layer.eval()
    pe = PositionEmbeddingCoordsSine(d_pos=d, pos_type="fourier", normalize=True)
    pe.load_state_dict(synthetic_state_dict(pe.state_dict(), 3))
    rng = np.random.default_rng(4)
    tgt = torch.from_numpy(rng.standard_normal((nq, B, d)).astype(np.float32))
    mem = torch.from_numpy(rng.standard_normal((nc, B, d)).astype(np.float32))
    qpos = torch.from_numpy(rng.standard_normal((nq, B, d)).astype(np.float32))
    g = torch.from_numpy(rng.uniform(0, 6, (B, nq * nc, 3)).astype(np.float32))
    lo = torch.from_numpy(rng.uniform(-3, -2, (B, 3)).astype(np.float32))
    hi = torch.from_numpy(rng.uniform(2, 3, (B, 3)).astype(np.float32))
    rel = pe(g, input_range=[hi, lo]).reshape(B, -1, nq, nc).permute(2, 3, 0, 1)
    with torch.no_grad():
        out, _ = layer(tgt, mem, query_pos=qpos, relative_pos=rel)
    np.savez_compressed(os.path.join(HERE, "decoder_layer.npz"), tgt=tgt.numpy(), memory=mem.numpy(),
                        query_pos=qpos.numpy(), geo=g.numpy(), lo=lo.numpy(), hi=hi.numpy(),
                        relative_pos_sub=rel.numpy()[::4, ::8], out=out.numpy())
    print("decoder golden ok", out.abs().mean().item())


def golden_matrix_nms():
    """matrix_non_max_suppression (util/utils_3d.py:95-141) on overlapping random proposals, both kernels."""
    from util.utils_3d import matrix_non_max_suppression

    rng = np.random.default_rng(11)
    n, N = 40, 3000
    base = rng.integers(0, N - 400, 12)
    masks = np.zeros((n, N), np.float32)
    for i in range(n):
        b = base[rng.integers(0, len(base))] + rng.integers(-60, 60)
        w = rng.integers(150, 400)
        masks[i, max(b, 0):b + w] = 1
        masks[i, rng.integers(0, N, 30)] = 1
    scores = rng.uniform(0.2, 1.0, n).astype(np.float32)
    cats = rng.integers(4, 8, n).astype(np.int64)
    out = {"masks": masks.astype(np.int32), "scores": scores, "categories": cats}
    for kern in ("gaussian", "linear"):
        for thr in (0.5, 0.05):
            pick = matrix_non_max_suppression(torch.from_numpy(masks), torch.from_numpy(scores), torch.from_numpy(cats),
                                              kernel=kern, final_score_thresh=thr)
            out[f"pick_{kern}_{thr}"] = pick.numpy()
    np.savez_compressed(os.path.join(HERE, "matrix_nms.npz"), **out)
    print("matrix_nms.npz", {k: v.shape for k, v in out.items()})


def fs_dicts():
    """Query scene + one full support scene with one labelled cuboid as support mask."""
    q = scene.make_batch([scene.make_small_scene(8192, 7)])
    sup_sc = scene.make_small_scene(6000, 8)
    sup = scene.make_batch([sup_sc])
    for d in (q, sup):
        d["batch_offsets"] = d["offsets"]
    sup["support_masks"] = (sup["instance_labels"] >= 0).long()
    return sup, q


def golden_fs():
    torch.manual_seed(0)
    m = GeoFormerFS()
    sd = m.state_dict()
    json.dump({k: list(v.shape) for k, v in sd.items()},
              open(os.path.join(HERE, "geoformer_fs_state_dict_keys.json"), "w"), indent=0)
    m.load_state_dict(synthetic_state_dict(sd, 2))
    m.semantic_linear.bias.data[3] += 1.5  # random head: lift class 3 (the test-fold foreground, geoformer_fs.py:466-469)
    m.eval()
    sup, q = fs_dicts()
    cap = {}
    orig_gmp = m.get_mask_prediction

    def gmp(*a, **k):
        r = orig_gmp(*a, **k)
        cap.setdefault("mask_logits", []).append(r[-1]["mask_logits"][0].detach().numpy().copy())
        return r

    m.get_mask_prediction = gmp
    orig_sim = m.similarity_net.forward
    with torch.no_grad():
        emb = m.process_support(sup, training=False)
        out = m(sup, q, training=False, remember=False, support_embeddings=None)
        out2 = m(sup, q, training=False, remember=True, support_embeddings=emb * 0.5)  # cached query side
    ctx = m.cache_data
    ml = None
    scores, props = out["proposal_scores"]
    scores2, props2 = out2["proposal_scores"]
    np.savez_compressed(
        os.path.join(HERE, "geoformer_fs_s8k_eval.npz"), weight_seed=2, semantic_bias3_shift=1.5,
        mask_logits_sub=sub(cap["mask_logits"][0]), mask_logits_half_sub=sub(cap["mask_logits"][1]),
        support_embeddings=emb.numpy(), semantic_scores=out["semantic_scores"].numpy(), fg_idxs=ctx[3].numpy(),
        pre_enc_inds=ctx[2].numpy(), context_feats=ctx[1].numpy(),
        proposal_scores=np.asarray(scores), proposal_npoints=np.asarray(props).sum(1) if len(scores) else np.zeros(0),
        proposal_scores_half=np.asarray(scores2),
        proposal_npoints_half=np.asarray(props2).sum(1) if len(scores2) else np.zeros(0))
    print("fs golden: N_fg", ctx[3].shape[0], "proposals", len(scores), len(scores2), "emb", emb.shape)


def train_case(mid):
    """Inputs of the training golden, shared with tests/test_training_golden.py through tests.util.train_golden_case."""
    from tests.util import train_golden_case

    return train_golden_case(mid)


def golden_train(mid):
    """The reference's GeoFormer.forward(batch, epoch > prepare_epochs, training=True) (geoformer.py:402-493) + its
    InstSetCriterion (criterion.py:137-245) + backward(), as train.py:63-75 runs them.  `self.get_batch_offsets` is
    undefined in the shipped class (geoformer.py:482); the instance gets util.utils.get_batch_offsets
    as the attribute, which is the intended function.  Dropout off everywhere."""
    import criterion as ref_crit  # reference module
    import model.geoformer.geoformer as ref_gf
    from util import utils as ref_utils
    from util.config import cfg

    case = train_case(mid)
    for k, v in case["cfg"].items():
        setattr(cfg, k, v)
    # the CPU `.to(device)` of a requires-grad leaf returns the leaf itself; on a GPU it is a copy the criterion then
    # updates in place (criterion.py:139,203) -- same patch as make_golden_criterion.py
    _to = torch.Tensor.to

    def _to_like_gpu(self, *a, **k):
        r = _to(self, *a, **k)
        return r.clone() if (r is self and self.requires_grad and self.is_leaf) else r

    torch.Tensor.to = _to_like_gpu
    torch.Tensor.cuda = lambda self, *a, **k: (self.clone() if (self.requires_grad and self.is_leaf) else self)
    if case["n_subsample"] != 30000:  # small case: cut the hard-coded 30 000 so np.random.choice is actually drawn
        orig_rd = ref_gf.random_downsample
        ref_gf.random_downsample = lambda bo, bs, n_subsample=30000: orig_rd(bo, bs, n_subsample=case["n_subsample"])
    torch.manual_seed(0)
    m = GeoFormer()
    m.load_state_dict(synthetic_state_dict(m.state_dict(), case["weight_seed"]))
    with torch.no_grad():
        m.semantic_linear.bias[4:] += case["fg_bias"]
    m.get_batch_offsets = ref_utils.get_batch_offsets
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.train()
    batch = case["batch"]()
    cap = {}
    orig_agg, orig_gmp = m.forward_aggregator, m.get_mask_prediction

    def agg(*a, **k):
        r = orig_agg(*a, **k)
        cap["pre_enc_inds"] = r[2].detach().numpy().copy()
        return r

    def gmp(geo, dec_outputs, *a, **k):
        cap["dec_outputs"] = dec_outputs.detach().numpy().copy()
        cap["geo_reached"] = np.stack([(g >= 0).sum(1).numpy() for g in geo])
        return orig_gmp(geo, dec_outputs, *a, **k)

    m.forward_aggregator, m.get_mask_prediction = agg, gmp
    np.random.seed(case["numpy_seed"])
    epoch = cfg.prepare_epochs + 1
    out = m(batch, epoch)
    crit = ref_crit.InstSetCriterion()
    loss, ld = crit(out, batch, epoch)
    m.zero_grad()
    loss.backward()
    res = {"loss": float(loss), "epoch": epoch, "n_layers": len(out["mask_predictions"])}
    for k, v in ld.items():
        res["ld_" + k] = np.array(v, np.float64)
    res["fg_idxs"] = out["fg_idxs"].numpy()  # = fg_idxs[idxs_subsample]: pins the subsample draw
    res["batch_idxs"] = out["batch_idxs"].numpy()
    res["semantic_scores_sub"] = np.ascontiguousarray(out["semantic_scores"].detach().numpy()[::16])
    res["pre_enc_inds"] = cap["pre_enc_inds"]
    res["dec_outputs"] = cap["dec_outputs"]
    res["geo_reached"] = cap["geo_reached"]
    for l, mp in enumerate(out["mask_predictions"]):
        res[f"cls_logits_{l}"] = mp["cls_logits"].detach().numpy()
        for b, ml in enumerate(mp["mask_logits"]):
            ml = ml.detach().numpy()
            res[f"mask_logits_sub_{l}_{b}"] = sub(ml, 4, 16 if mid else 4)
            res[f"mask_logits_rowsum_{l}_{b}"] = ml.astype(np.float64).sum(1)
    names, gnorm, gsum, gsamp, goff = [], [], [], [], [0]
    for n, p in m.named_parameters():
        g = np.zeros(p.shape, np.float32) if p.grad is None else p.grad.numpy()
        g64 = g.astype(np.float64).ravel()
        names.append(n + ("" if p.grad is not None else "|none"))
        gnorm.append(np.sqrt((g64 ** 2).sum()))
        gsum.append(g64.sum())
        stride = max(1, g64.size // 256)
        gsamp.append(g.ravel()[::stride].copy())
        goff.append(goff[-1] + gsamp[-1].size)
    res["grad_names"] = np.array(names)
    res["grad_norm"], res["grad_sum"] = np.array(gnorm), np.array(gsum)
    res["grad_samples"], res["grad_sample_offsets"] = np.concatenate(gsamp), np.array(goff)
    f = os.path.join(HERE, "geoformer_train_mid.npz" if mid else "geoformer_train_small.npz")
    np.savez_compressed(f, **res)
    print("train golden", "mid" if mid else "small", "loss", res["loss"], {k: np.asarray(v).ravel()[:2].tolist() for k, v in res.items() if k.startswith("ld_")},
          "N_sub", res["fg_idxs"].shape, "size", os.path.getsize(f))


if __name__ == "__main__":
    if _TRAIN:
        golden_train(_WHICH == ["train_mid"])
        sys.exit(0)
    which = _WHICH or ["geodesic", "decoder", "model"]
    if "fs" in which:
        golden_fs()
    if "nms" in which:
        golden_matrix_nms()
    if "geodesic" in which:
        golden_geodesic()
    if "decoder" in which:
        golden_decoder_layer()
    if "model" in which:
        golden_model()
