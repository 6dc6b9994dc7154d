"""The build's training criteria against fixtures produced by the REFERENCE's own criterion.py / criterion_fs.py /
model/matcher.py (tests/golden/make_golden_criterion.py: imported from /root/reference in the build container).

Inputs are regenerated from the stored seeds (tests.util.criterion_case, shared with the generator); compared are
the total loss and every reported component, the Hungarian assignment (matched query rows, and the matched
ground-truth instance identified by its size, first point and class), and the gradient sent back into every model
output (strided sample + l2 norm + sum).  Runs on the CPU, and on the GPU when there is one."""
import os

import numpy as np
import pytest
import torch

from tests.util import criterion_case

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run(crit, case, epoch, fs, dev, B, NQ, NL):
    t = lambda a, g=False: torch.from_numpy(a).to(dev).clone().requires_grad_(g)  # noqa: E731
    sem = t(case["semantic_scores"], True)
    cls = [t(case["cls_logits"][l], True) for l in range(NL)]
    ml = [[t(case[f"mask_logits_{l}_{b}"], True) for b in range(B)] for l in range(NL)]
    outputs = {"semantic_scores": sem, "fg_idxs": t(case["fg_idxs"]), "batch_idxs": t(case["batch_idxs"]),
               "mask_predictions": [{"cls_logits": cls[l], "mask_logits": ml[l]} for l in range(NL)]}
    sim = None
    if fs:
        sim = t(case["simnet"], True)
        outputs["simnet"] = sim
    batch = {"labels": t(case["labels"]), "instance_labels": t(case["instance_labels"])}
    loss, ld = crit(outputs, batch, epoch)
    loss.backward()
    return loss, ld, sem, cls, ml, sim


def _check(z, name, loss, ld, crit, case, sem, cls, ml, sim, B, NL, fs):
    pre = name + "_out_"
    assert abs(float(loss) - float(z[pre + "loss"])) < 1e-5 * max(1.0, abs(float(z[pre + "loss"])))
    ref_ld = {k[len(pre) + 3:]: z[k] for k in z.files if k.startswith(pre + "ld_")}
    assert set(ld) == set(ref_ld)
    for k, v in ld.items():
        assert abs(v[0] - ref_ld[k][0]) < 1e-5 * max(1.0, abs(ref_ld[k][0])) and v[1] == ref_ld[k][1], k
    for b in range(B):
        if pre + f"match_rows_{b}" not in z.files:
            continue
        rows, gm, sc = crit.cached[b]
        assert (np.asarray(rows) == z[pre + f"match_rows_{b}"]).all()  # same assignment
        assert (gm.sum(1).cpu().numpy() == z[pre + f"match_gt_npoints_{b}"]).all()
        first = np.array([int(torch.nonzero(r)[0]) for r in gm])
        assert (first == z[pre + f"match_gt_first_{b}"]).all()
        assert (sc.cpu().numpy() == z[pre + f"match_sem_{b}"]).all()

    def grad_ok(key, g, like):
        g = np.zeros_like(like) if g is None else g.detach().cpu().numpy()
        assert np.abs(g[..., ::3, ::7] - z[pre + "gsub_" + key]).max() < 1e-6, key
        n = z[pre + "gnorm_" + key]
        assert abs(np.sqrt((g.astype(np.float64) ** 2).sum()) - n[0]) < 1e-5 * max(1e-3, n[0]), key
        assert abs(g.astype(np.float64).sum() - n[1]) < 1e-5 * max(1e-2, abs(n[1]), n[0]), key

    grad_ok("semantic_scores", sem.grad, case["semantic_scores"])
    g_cls = np.stack([np.zeros_like(case["cls_logits"][0]) if c.grad is None else c.grad.cpu().numpy() for c in cls])
    grad_ok("cls_logits", torch.from_numpy(g_cls), case["cls_logits"])
    for l in range(NL):
        for b in range(B):
            grad_ok(f"mask_logits_{l}_{b}", ml[l][b].grad, case[f"mask_logits_{l}_{b}"])
    if fs:
        g = np.zeros_like(case["simnet"]) if sim.grad is None else sim.grad.cpu().numpy()
        assert np.abs(g - z[pre + "grad_simnet"]).max() < 1e-6


def _std(dev):
    from geoformer_amd.model import load_config
    from geoformer_amd.model.criterion import InstSetCriterion

    z = np.load(os.path.join(G, "criterion_std.npz"))
    B, NQ, NL = int(z["B"]), int(z["NQ"]), int(z["NL"])
    cfg = load_config("geoformer_scannet.yaml", batch_size=B, n_query_points=NQ, dec_nlayers=NL)
    assert cfg.prepare_epochs == int(z["prepare_epochs"])
    crit = InstSetCriterion(cfg)
    for name in ("a", "b", "pre"):
        case = criterion_case(int(z[name + "_seed"]), False, B, NQ, NL)
        loss, ld, sem, cls, ml, sim = _run(crit, case, int(z[name + "_epoch"]), False, dev, B, NQ, NL)
        _check(z, name, loss, ld, crit, case, sem, cls, ml, sim, B, NL, False)


def _fs(dev):
    from geoformer_amd.model import load_config
    from geoformer_amd.model.criterion_fs import FSInstSetCriterion

    z = np.load(os.path.join(G, "criterion_fs.npz"))
    B, NQ, NL = int(z["B"]), int(z["NQ"]), int(z["NL"])
    cfg = load_config("geoformer_fs_scannet.yaml", batch_size=B, n_query_points=NQ, dec_nlayers=NL)
    assert cfg.negative_ratio == int(z["negative_ratio"]) and list(cfg.fix_module) == list(z["fix_module"])
    crit = FSInstSetCriterion(cfg)
    for name in ("a", "b"):
        case = criterion_case(int(z[name + "_seed"]), True, B, NQ, NL)
        loss, ld, sem, cls, ml, sim = _run(crit, case, int(z[name + "_epoch"]), True, dev, B, NQ, NL)
        _check(z, name, loss, ld, crit, case, sem, cls, ml, sim, B, NL, True)


def test_inst_set_criterion_matches_reference():
    _std("cpu")


def test_fs_inst_set_criterion_matches_reference():
    _fs("cpu")


@pytest.mark.gpu
def test_criteria_match_reference_on_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _std("cuda")
    _fs("cuda")
