"""CPU: the build's own GeoFormer (host logic, index-space quirks, numpy RNG consumption, torch
modules) driven through the oracle-backed operator stand-ins, against the golden produced by the
REFERENCE's forward().  Pins everything above the native layer without a GPU."""
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_geoformer_forward_cpu_matches_reference_golden(oracle):
    from geoformer_amd import scene
    from geoformer_amd.model import GeoFormer, load_config
    from oracle import cpu_backend
    from tests.util import synthetic_state_dict

    z = np.load(os.path.join(G, "geoformer_s8k_eval.npz"))
    m = GeoFormer(load_config("test_geoformer_scannet.yaml"))
    m.load_state_dict(synthetic_state_dict(m.state_dict(), int(z["weight_seed"])))
    m.eval()
    batch = scene.make_batch([scene.make_small_scene(int(z["scene_points"]), int(z["scene_seed"]))])
    np.random.seed(int(z["numpy_seed"]))
    with cpu_backend.installed(), torch.no_grad():
        out = m(batch, 300, training=False)
    assert np.abs(out["semantic_scores"].numpy() - z["semantic_scores"]).max() < 1e-4
    assert (out["fg_idxs"].numpy() == z["fg_idxs"]).all()
    assert (m.last_sampling_indices.numpy() == z["sampling_indices"]).all()
    mp = out["mask_predictions"][-1]
    assert np.abs(mp["cls_logits"].numpy() - z["cls_logits"]).max() < 1e-4
    ml = mp["mask_logits"][0].numpy()
    assert np.abs(ml[::8, ::4] - z["mask_logits_sub"]).max() < 1e-4
    cls_final, scores_final, masks_final = out["proposal_scores"]
    assert (cls_final.numpy() == z["proposal_cls"]).all()
    assert np.abs(scores_final.numpy() - z["proposal_scores"]).max() < 1e-4
    # early-exit branches of forward(): backbone-only epochs and an empty foreground set
    with cpu_backend.installed(), torch.no_grad():
        early = m(batch, m.prepare_epochs, training=False)
        assert set(early) == {"semantic_scores"}
        m.semantic_linear.bias.data[:4] += 1e4  # nothing is predicted as an object class
        empty = m(batch, 300, training=False)
    assert empty["mask_predictions"] is None


def test_geoformer_fs_state_dict_and_episode_cpu(oracle):
    """GeoFormerFS: parameter names/shapes and a 1-shot episode (support branch, fusion, similarity net,
    `remember` cache) against the reference's GeoFormerFS run on CPU."""
    import json

    from geoformer_amd.model import GeoFormerFS, load_config
    from oracle import cpu_backend
    from tests.util import check_fs_episode, run_fs_episode

    ref = json.load(open(os.path.join(G, "geoformer_fs_state_dict_keys.json")))
    m = GeoFormerFS(load_config("geoformer_fs_scannet.yaml"))
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert mine == ref or (set(mine) == set(ref) and all(mine[k] == ref[k] for k in ref))
    assert sum(p.numel() for p in m.parameters()) == 8142687
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 42706
    with cpu_backend.installed():
        check_fs_episode(*run_fs_episode("cpu"))
