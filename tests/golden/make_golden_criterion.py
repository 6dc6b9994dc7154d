"""Golden fixtures for the training criteria, produced by the REFERENCE's own criterion.py / criterion_fs.py /
model/matcher.py (imported from /root/reference in the build container; nothing is copied).

    python tests/golden/make_golden_criterion.py std    -> criterion_std.npz  (InstSetCriterion, train yaml)
    python tests/golden/make_golden_criterion.py fs     -> criterion_fs.npz   (FSInstSetCriterion, few-shot yaml)

The criteria are pure functions of (model_outputs, batch_inputs, epoch): the inputs are seeded synthetic tensors of
the shapes GeoFormer.forward(training=True) emits, stored in the fixture together with the reference's
losses, its Hungarian assignment and the gradients it sends back into every model output.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
MODE = sys.argv[1] if len(sys.argv) > 1 else "std"
_YAML = "config/geoformer_fs_scannet.yaml" if MODE == "fs" else "config/geoformer_scannet.yaml"
sys.argv = ["make_golden_criterion", "--config", os.path.join(REF, _YAML)]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.golden import ref_shims  # noqa: E402

ref_shims.install(REF)
os.chdir(REF)
# `torch.tensor(0.0, requires_grad=True).to(device)` (criterion.py:139,203) is a COPY of the leaf on a GPU, which the
# reference then updates in place; on the CPU `.to` returns the leaf itself and autograd refuses the `+=`.  Give the
# CPU run the GPU's behaviour: a differentiable copy.
_to = torch.Tensor.to


def _to_like_gpu(self, *a, **k):
    r = _to(self, *a, **k)
    return r.clone() if (r is self and self.requires_grad and self.is_leaf) else r


torch.Tensor.to = _to_like_gpu
torch.Tensor.cuda = lambda self, *a, **k: (self.clone() if (self.requires_grad and self.is_leaf) else self)
from util.config import cfg  # noqa: E402  (reference config namespace)

B, NQ, NL, NCLS = 3, 32, 4, 13
cfg.batch_size, cfg.n_query_points, cfg.dec_nlayers = B, NQ, NL


from tests.util import criterion_case as synthetic_case  # noqa: E402  (shared with the tests: inputs are not stored)


def run(crit, case, epoch, fs):
    t = lambda a, g=False: torch.from_numpy(a).clone().requires_grad_(g)  # noqa: E731
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
    res = {"loss": float(loss)}
    for k, v in ld.items():
        res["ld_" + k] = np.array(v, np.float64)
    for b, (pi, gm, sc) in enumerate(crit.cached):
        res[f"match_rows_{b}"] = np.asarray(pi, np.int64)
        res[f"match_gt_npoints_{b}"] = gm.sum(1).numpy()
        res[f"match_gt_first_{b}"] = np.array([int(torch.nonzero(r)[0]) for r in gm], np.int64)
        res[f"match_sem_{b}"] = sc.numpy()
    z = lambda g, like: np.zeros_like(like) if g is None else g.numpy()  # noqa: E731

    def put(name, g):  # gradients: a strided sample + the exact l2 norm and sum (float64)
        res["gsub_" + name] = np.ascontiguousarray(g[..., ::3, ::7])
        res["gnorm_" + name] = np.array([np.sqrt((g.astype(np.float64) ** 2).sum()), g.astype(np.float64).sum()])

    put("semantic_scores", z(sem.grad, case["semantic_scores"]))
    put("cls_logits", np.stack([z(c.grad, case["cls_logits"][0]) for c in cls]))
    for l in range(NL):
        for b in range(B):
            put(f"mask_logits_{l}_{b}", z(ml[l][b].grad, case[f"mask_logits_{l}_{b}"]))
    if fs:
        res["grad_simnet"] = z(sim.grad, case["simnet"])
    return res


if MODE == "fs":
    import criterion_fs as ref_crit  # noqa: E402  (reference module)

    # torch.LongTensor(range(n)) ... .cuda() and friends are CPU no-ops under ref_shims
    out = {"B": B, "NQ": NQ, "NL": NL, "negative_ratio": cfg.negative_ratio, "fix_module": np.array(list(cfg.fix_module))}
    crit = ref_crit.FSInstSetCriterion()
    for name, seed, epoch in (("a", 11, 5), ("b", 12, 5)):
        case = synthetic_case(seed, True, B, NQ, NL, NCLS)
        res = run(crit, case, epoch, True)
        out.update({f"{name}_out_{k}": v for k, v in res.items()})
        out[f"{name}_epoch"], out[f"{name}_seed"] = epoch, seed
        print("fs", name, "loss", res["loss"], {k: v for k, v in res.items() if k.startswith("ld_")})
    np.savez_compressed(os.path.join(HERE, "criterion_fs.npz"), **out)
else:
    import criterion as ref_crit  # noqa: E402  (reference module)

    out = {"B": B, "NQ": NQ, "NL": NL, "prepare_epochs": cfg.prepare_epochs}
    crit = ref_crit.InstSetCriterion()
    for name, seed, epoch in (("a", 1, cfg.prepare_epochs + 1), ("b", 2, cfg.prepare_epochs + 50), ("pre", 3, 1)):
        case = synthetic_case(seed, False, B, NQ, NL, NCLS)
        crit.cached = []
        res = run(crit, case, epoch, False)
        out.update({f"{name}_out_{k}": v for k, v in res.items()})
        out[f"{name}_epoch"], out[f"{name}_seed"] = epoch, seed
        print("std", name, "loss", res["loss"], {k: v for k, v in res.items() if k.startswith("ld_")})
    np.savez_compressed(os.path.join(HERE, "criterion_std.npz"), **out)
