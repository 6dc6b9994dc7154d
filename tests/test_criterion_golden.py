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
        cached = crit.matches_reference_format if getattr(crit, "device_matches", None) else crit.cached
        rows, gm, sc = cached[b]
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
@pytest.mark.parametrize("route", ["device", "host"])
def test_criteria_match_reference_on_gpu(route, monkeypatch):
    """device: matching by gf_lsap and masked losses without per-scene read-backs (csrc/lsap.hip, criterion.py);
    host: the scipy route on CUDA tensors.  Both against the reference's fixtures."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    monkeypatch.setenv("GF_DEVICE_CRITERION", "1" if route == "device" else "0")
    _std("cuda")
    _fs("cuda")


@pytest.mark.gpu
def test_lsap_equals_scipy():
    """gf_lsap against scipy.optimize.linear_sum_assignment on the same fp32 matrices: more queries than instances
    (the training shape, transposed inside), more instances than queries, square, absent instances, and integer-valued
    costs full of ties (where the scan order and the tie-breaking rule decide the assignment)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from scipy.optimize import linear_sum_assignment

    from geoformer_amd import _lib
    from geoformer_amd._lib import check, ptr, stream_ptr

    lib = _lib.load()
    rng = np.random.default_rng(0)
    cases = [(128, 7, 0), (128, 40, 5), (32, 32, 0), (16, 50, 3), (256, 1, 0), (128, 64, 64), (8, 300, 17), (300, 9, 2)]
    for nq, K, absent in cases:
        for ties in (False, True):
            cost = rng.integers(0, 4, (nq, K)).astype(np.float32) if ties else rng.standard_normal((nq, K)).astype(np.float32)
            present = np.ones(K, np.int32)
            if absent:
                present[rng.choice(K, min(absent, K), replace=False)] = 0
            c = torch.from_numpy(cost).cuda()
            pr = torch.from_numpy(present).cuda()
            mq = torch.empty(K, dtype=torch.int32, device="cuda")
            moq = torch.empty(nq, dtype=torch.int32, device="cuda")
            nm = torch.empty(1, dtype=torch.int32, device="cuda")
            st = torch.empty(1, dtype=torch.int32, device="cuda")
            check(lib.gf_lsap(ptr(c), nq, K, ptr(pr), ptr(mq), ptr(moq), ptr(nm), ptr(st), stream_ptr()), "gf_lsap")
            assert int(st) == 0
            cols_present = np.nonzero(present)[0]
            if cols_present.size == 0:
                assert int(nm) == 0 and (mq.cpu().numpy() == -1).all()
                continue
            rows, cols = linear_sum_assignment(cost[:, cols_present])
            ref = np.full(nq, -1, np.int64)
            ref[rows] = cols_present[cols]
            got = moq.cpu().numpy()
            assert (got == ref).all(), (nq, K, absent, ties)
            assert int(nm) == rows.size
            inv = np.full(K, -1, np.int64)
            inv[cols_present[cols]] = rows
            assert (mq.cpu().numpy() == inv).all()


@pytest.mark.gpu
def test_pair_losses_kernel_equals_operator_formulation():
    """gf_pair_losses_fwd / _bwd (the fused dice + focal loss of a scene's matched pairs) against the operator-by-operator
    PyTorch formulation of criterion.masked_pair_losses on the same device match: values and the gradient with respect
    to the mask logits, with absent instances, unmatched queries, more instances than queries, and large logits."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from geoformer_amd.model import criterion as C

    g = torch.Generator().manual_seed(3)
    for nq, K, n, absent, scale in ((128, 9, 30000, 2, 3.0), (16, 40, 5001, 0, 1.0), (64, 1, 777, 0, 20.0), (8, 5, 64, 5, 1.0)):
        x = (torch.randn(nq, n, generator=g) * scale).cuda().requires_grad_(True)
        ids = torch.randint(0, K, (n,), generator=g)
        present = torch.ones(K, dtype=torch.bool)
        present[torch.randperm(K, generator=g)[:absent]] = False
        ids[~present[ids]] = -100 + 0 * ids[~present[ids]]
        m = C.DeviceMatch()
        m.lo = 0
        m.inst_masks = (ids[None, :] == torch.arange(K)[:, None]).float().cuda().contiguous()
        pres = (m.inst_masks.sum(1) > 0).cpu()
        cols = torch.nonzero(pres).flatten()
        nm = min(nq, cols.numel())
        qs = torch.randperm(nq, generator=g)[:nm]
        ks = cols[torch.randperm(cols.numel(), generator=g)[:nm]]
        mq = torch.full((K,), -1, dtype=torch.int32)
        moq = torch.full((nq,), -1, dtype=torch.int32)
        mq[ks] = qs.int()
        moq[qs] = ks.int()
        m.match_q, m.match_of_q = mq.cuda(), moq.cuda()
        m.n_match = torch.tensor([nm], dtype=torch.int32).cuda()
        w = torch.tensor([0.7, 1.3]).cuda()
        n_f = m.n_match[0].float()
        # reference: the operator formulation (the same function, taken on a float64 copy it does not fuse)
        xr = x.detach().double().requires_grad_(True)
        m64 = C.DeviceMatch()
        m64.inst_masks, m64.match_q, m64.match_of_q, m64.n_match = m.inst_masks.double(), m.match_q, m.match_of_q, m.n_match
        dr, fr = C.masked_pair_losses(xr, m64, n_f.double())
        (dr * w[0] + fr * w[1]).backward()
        dg, fg = C.masked_pair_losses(x, m, n_f)
        (dg * w[0] + fg * w[1]).backward()
        assert abs(dg.item() - dr.item()) <= 1e-5 * max(1.0, abs(dr.item())), (nq, K, n)
        assert abs(fg.item() - fr.item()) <= 1e-5 * max(1.0, abs(fr.item())), (nq, K, n)
        gr = xr.grad.float()
        assert (x.grad - gr).abs().max().item() <= 1e-5 * max(gr.abs().max().item(), 1e-12), (nq, K, n)
        if nm:
            unmatched = (m.match_of_q < 0)
            assert (x.grad[unmatched] == 0).all()
