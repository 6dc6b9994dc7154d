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
