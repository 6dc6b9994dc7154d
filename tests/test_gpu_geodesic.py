"""GPU parity of the geodesic stage: radius-limited kNN rows and BFS distances vs the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _pts(n, seed):
    from geoformer_amd import scene

    sc = scene.make_scene(max(n, 64), seed)
    p = sc["xyz"]
    return np.ascontiguousarray(p[np.random.default_rng(seed).permutation(p.shape[0])[:n]])


def _ref_graph(oracle, xyz, k, radius):
    D2, I = oracle.knn(xyz, xyz, k)
    D = np.sqrt(D2)
    return D, I


@pytest.mark.parametrize("n", [6000, 20000])
def test_knn_radius_matches_bruteforce(hip, oracle, n):
    from geoformer_amd import pointops

    xyz = _pts(n, 17 + n)
    xyz[50:60] = xyz[3]  # duplicates: ties on d2 resolved by index
    k, radius = 64, 0.05
    D, I = _ref_graph(oracle, xyz, k, radius)
    gd, gi, deg = pointops.knn_radius(_dev(xyz), k, radius, sqrt_out=True, check_overflow=True)
    gd, gi, deg = gd.cpu().numpy(), gi.cpu().numpy(), deg.cpu().numpy()
    inr = D <= np.float32(radius)
    # within the radius the rows agree entry by entry (index bit-exact, distance bit-exact)
    assert (np.where(inr, I, -1) == gi).all()
    assert (np.where(inr, D, np.inf) == gd).all()
    assert (deg == inr.sum(1) - 1).all()


@pytest.mark.parametrize("n,nq,max_step", [(6000, 16, 256), (20000, 64, 128), (20000, 8, 5)])
def test_bfs_matches_oracle(hip, oracle, n, nq, max_step):
    from geoformer_amd import pointops

    xyz = _pts(n, 5 + n)
    k, radius = 64, 0.05
    D, I = _ref_graph(oracle, xyz, k, radius)
    rng = np.random.default_rng(1)
    src = rng.integers(0, n, nq)
    ref = oracle.geodesic(D[:, 1:], I[:, 1:], src, radius, max_step)
    gd, gi, deg = pointops.knn_radius(_dev(xyz), k, radius)
    geo = pointops.geodesic_bfs(gd, gi, deg, _dev(src.astype(np.int32)), radius, max_step).cpu().numpy()
    assert ((geo >= 0) == (ref >= 0)).all()  # reach sets bit-exact
    assert (geo == ref).all()  # fp32 sums along the same parent chain -> bit-exact
    # also through a full (not radius-limited) table without the degree shortcut
    geo2 = pointops.geodesic_bfs(_dev(D.astype(np.float32)), _dev(I.astype(np.int32)), None,
                                 _dev(src.astype(np.int32)), radius, max_step).cpu().numpy()
    assert (geo2 == ref).all()
    # several queries per compute unit (the layout the forward uses beside furthest point sampling): same values
    for wg in (256, 512):
        geo3 = pointops.geodesic_bfs(gd, gi, deg, _dev(src.astype(np.int32)), radius, max_step, wg_threads=wg)
        assert (geo3.cpu().numpy() == ref).all()


def test_bfs_duplicates_short_walks_and_dense_ball(hip, oracle):
    """Corner cases of the frontier BFS against the oracle: 100 copies of one point (rows of 63 duplicates at distance 0,
    a frontier that jumps by 100 vertices at once), hop limits of 1 / 2 / 7, and a 0.06 m ball of 3000 points whose
    frontier reaches ~1900 vertices in the second hop (table from the oracle's brute force, laid out like
    gf_knn_radius does: that kernel caps its candidate list at 1024 in-radius points)."""
    from geoformer_amd import pointops

    n, k, radius = 20000, 64, 0.05
    for dup in (False, True):
        xyz = _pts(n, 77)
        if dup:
            xyz[1000:1100] = xyz[5]
        D, I = _ref_graph(oracle, xyz, k, radius)
        src = np.random.default_rng(2).integers(0, n, 24)
        if dup:
            src[:3] = [5, 1003, 1099]
        ref = oracle.geodesic(D[:, 1:], I[:, 1:], src, radius, 256)
        gd, gi, deg = pointops.knn_radius(_dev(xyz), k, radius)
        for wg in (256, 1024):
            geo = pointops.geodesic_bfs(gd, gi, deg, _dev(src.astype(np.int32)), radius, 256, wg_threads=wg)
            assert (geo.cpu().numpy() == ref).all(), (dup, wg)
    for max_step in (1, 2, 7):
        ref = oracle.geodesic(D[:, 1:], I[:, 1:], src, radius, max_step)
        geo = pointops.geodesic_bfs(gd, gi, deg, _dev(src.astype(np.int32)), radius, max_step, wg_threads=256)
        assert (geo.cpu().numpy() == ref).all(), max_step
    rng = np.random.default_rng(9)
    xyz = (rng.random((3000, 3)) * 0.06).astype(np.float32)
    D, I = _ref_graph(oracle, xyz, k, radius)
    inr = D <= np.float32(radius)
    Dm = np.where(inr, D, np.inf).astype(np.float32)
    Im = np.where(inr, I, -1).astype(np.int32)
    src = np.array([0, 17, 2999])
    ref = oracle.geodesic(D[:, 1:], I[:, 1:], src, radius, 64)
    deg = (inr.sum(1) - 1).astype(np.int32)
    geo = pointops.geodesic_bfs(_dev(Dm), _dev(Im), _dev(deg), _dev(src.astype(np.int32)), radius, 64, wg_threads=256)
    assert (geo.cpu().numpy() == ref).all()
    # 9000 points in a 0.09 m cube: every row has all 64 entries inside the radius (the lanes go through all four
    # 16-entry rounds of a row), rings of up to 1800 vertices
    xyz = (rng.random((9000, 3)) * 0.09).astype(np.float32)
    D, I = _ref_graph(oracle, xyz, k, radius)
    inr = D <= np.float32(radius)
    Dm = np.where(inr, D, np.inf).astype(np.float32)
    Im = np.where(inr, I, -1).astype(np.int32)
    src = np.array([3, 4500, 8999, 77])
    ref = oracle.geodesic(D[:, 1:], I[:, 1:], src, radius, 64)
    deg = (inr.sum(1) - 1).astype(np.int32)
    for wg in (256, 512, 1024):
        geo = pointops.geodesic_bfs(_dev(Dm), _dev(Im), _dev(deg), _dev(src.astype(np.int32)), radius, 64, wg_threads=wg)
        assert (geo.cpu().numpy() == ref).all(), wg
    # the same cube as vertices 250000.. of a graph of 2^19 vertices (the others isolated): the two bitmaps take 128 KB
    # of the workgroup's LDS, 1408 queue entries remain, and the rings above that spill into the queue's
    # global-memory overflow
    n_big, off = 1 << 19, 250000
    Db = np.full((n_big, k), np.inf, np.float32)
    Ib = np.full((n_big, k), -1, np.int32)
    Db[off:off + 9000] = Dm
    Ib[off:off + 9000] = np.where(Im >= 0, Im + off, -1)
    geo = pointops.geodesic_bfs(_dev(Db), _dev(Ib), None, _dev((src + off).astype(np.int32)), radius, 64, wg_threads=256)
    geo = geo.cpu().numpy()
    assert (geo[:, off:off + 9000] == ref).all()
    assert (geo[:, :off] == -1).all() and (geo[:, off + 9000:] == -1).all()
