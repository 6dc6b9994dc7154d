"""GPU: fused mask head vs a float64 evaluation of the reference formula (geoformer.py:286-324)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,nq,use_geo", [(5000, 37, True), (70, 256, True), (1234, 8, False)])
def test_mask_head_fused(hip, N, nq, use_geo):
    from geoformer_amd import pointops

    rng = np.random.default_rng(N + nq)
    feat = rng.standard_normal((N, 16)).astype(np.float32)
    coords = rng.uniform(-3, 3, (N, 3)).astype(np.float32)
    qxyz = coords[rng.integers(0, N, nq)].copy()
    geo = rng.uniform(0, 5, (nq, N)).astype(np.float32)
    geo[rng.uniform(size=geo.shape) < 0.3] = -1.0
    geo[0] = -1.0  # a query that reaches nothing takes the global maximum
    w1 = (rng.standard_normal((nq, 16, 19)) * 0.3).astype(np.float32)
    b1 = rng.standard_normal((nq, 16)).astype(np.float32)
    w2 = (rng.standard_normal((nq, 16)) * 0.3).astype(np.float32)
    b2 = rng.standard_normal(nq).astype(np.float32)
    rel = qxyz[:, None, :].astype(np.float64) - coords[None].astype(np.float64)
    mx = None
    if use_geo:
        m = geo.max(1)
        m = np.sqrt(np.where(m < 0, m.max(), m)).astype(np.float32)
        mx = m
        rel = np.where((geo < 0)[..., None], rel + m[:, None, None].astype(np.float64) * np.sign(rel), rel)
    x = np.concatenate([rel, np.broadcast_to(feat[None].astype(np.float64), (nq, N, 16))], 2)  # nq N 19
    h = np.maximum(np.einsum("qck,qnk->qnc", w1.astype(np.float64), x) + b1[:, None, :], 0)
    ref = np.einsum("qc,qnc->qn", w2.astype(np.float64), h) + b2[:, None]
    d = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    out = pointops.mask_head(d(feat), d(coords), d(geo) if use_geo else None, d(qxyz), d(mx), d(w1), d(b1), d(w2),
                             d(b2)).cpu().numpy()
    assert np.abs(out - ref).max() < 1e-4
