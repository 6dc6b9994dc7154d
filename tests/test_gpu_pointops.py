"""GPU parity of the point-set operators against the oracle (bit-exact indices, exact copies)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _scene_points(n, seed):
    from geoformer_amd import scene

    sc = scene.make_scene(max(n, 64), seed)
    p = sc["xyz"]
    rng = np.random.default_rng(seed)
    sel = rng.permutation(p.shape[0])[:n]
    return np.ascontiguousarray(p[sel])


def test_voxelize_fp_bp(hip, oracle):
    from geoformer_amd import pointops, scene

    sc = scene.make_small_scene(8192, 3)
    batch = scene.make_batch([sc])
    feats = np.concatenate([sc["rgb"], sc["xyz"]], 1).astype(np.float32)
    rules = batch["v2p_map"].numpy()
    ref = oracle.voxelize_fp(feats, rules, True)
    out = pointops.voxelize_fp(_dev(feats), _dev(rules), 4)
    assert (out.cpu().numpy() == ref).all()  # same rounding sequence -> bit-exact
    g = np.random.default_rng(0).standard_normal(ref.shape).astype(np.float32)
    refb = oracle.voxelize_bp(g, rules, feats.shape[0], True)
    d = torch.zeros(feats.shape, device="cuda")
    pointops.voxelize_bp(_dev(g), _dev(rules), 4, d)
    assert (d.cpu().numpy() == refb).all()


@pytest.mark.parametrize("n,m", [(20000, 2048), (50000, 2048), (3000, 32), (700, 2048), (5, 16), (513, 64)])
def test_fps_bit_exact(hip, oracle, n, m):
    from geoformer_amd import pointops

    xyz = _scene_points(n, n)[None]
    ref = oracle.fps(xyz, m)
    got = pointops.furthest_point_sampling(_dev(xyz), m).cpu().numpy()
    assert (got == ref).all()


def test_fps_ties_skip_and_batch(hip, oracle):
    from geoformer_amd import pointops

    rng = np.random.default_rng(11)
    a = _scene_points(4000, 5)
    a[100:140] = a[7]  # exact duplicates -> exact distance ties
    a[200] = 0.0  # |p|^2 <= 1e-3 is never selected (sampling_gpu.cu:104)
    a[201] = [0.02, 0.01, 0.0]
    b = np.round(rng.uniform(-1, 1, (4000, 3)) * 4) / 4  # lattice: many exact ties
    xyz = np.stack([a, b.astype(np.float32)])
    ref = oracle.fps(xyz, 300)
    got = pointops.furthest_point_sampling(_dev(xyz), 300).cpu().numpy()
    assert (got == ref).all()
    assert 200 not in got[0, 1:] and 201 not in got[0, 1:]


@pytest.mark.parametrize("n,m,ns", [(20000, 2048, 64), (50000, 2048, 64), (900, 37, 16)])
def test_ball_query_bit_exact(hip, oracle, n, m, ns):
    from geoformer_amd import pointops

    xyz = _scene_points(n, n + 1)[None]
    centers = xyz[:, oracle.fps(xyz, m)[0]]
    centers[0, -1] = 50.0  # a centre with no neighbour: row stays all-zero (ball_query.cpp:22-24)
    ref = oracle.ball_query(centers, xyz, 0.2, ns)
    for grid in (False, True):  # linear scan and hash-grid kernels: the same rows
        got = pointops.ball_query(_dev(centers), _dev(xyz), 0.2, ns, grid=grid).cpu().numpy()
        assert (got == ref).all(), grid
        assert (got[0, -1] == 0).all()
    # a radius with more hits than the grid kernel's candidate list holds: it falls back to the scan per centre
    big = oracle.ball_query(centers[:, :40], xyz, 1.5, ns)
    got = pointops.ball_query(_dev(centers[:, :40].copy()), _dev(xyz), 1.5, ns, grid=True).cpu().numpy()
    assert (got == big).all()


def test_gather_group_and_grads(hip, oracle):
    from geoformer_amd import pointops

    rng = np.random.default_rng(3)
    b, c, n, m, ns = 2, 16, 5000, 256, 64
    pts = rng.standard_normal((b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, (b, m)).astype(np.int32)
    gidx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    assert (pointops.gather_points(_dev(pts), _dev(idx)).cpu().numpy() == oracle.gather_points(pts, idx)).all()
    assert (pointops.group_points(_dev(pts), _dev(gidx)).cpu().numpy() == oracle.group_points(pts, gidx)).all()
    go = rng.standard_normal((b, c, m)).astype(np.float32)
    ggo = rng.standard_normal((b, c, m, ns)).astype(np.float32)
    r1 = oracle.gather_points_grad(go, idx, n)
    r2 = oracle.group_points_grad(ggo, gidx, n)
    assert np.abs(pointops.gather_points_grad(_dev(go), _dev(idx), n).cpu().numpy() - r1).max() < 1e-5
    assert np.abs(pointops.group_points_grad(_dev(ggo), _dev(gidx), n).cpu().numpy() - r2).max() < 1e-4


@pytest.mark.parametrize("n,m0,m", [(5000, 256, 2048), (700, 1, 64), (300, 100, 512), (4096, 33, 34)])
def test_fps_resume_equals_single_call(hip, n, m0, m):
    """gf_furthest_point_sampling_resume: continuing from the first m0 picks gives the sequence of one call."""
    from geoformer_amd import pointops

    rng = np.random.default_rng(n + m0)
    xyz = rng.uniform(-2, 2, (2, n, 3)).astype(np.float32)
    xyz[0, 10:20] = xyz[0, 5]  # duplicates -> exact ties
    x = torch.from_numpy(xyz).cuda()
    full = pointops.furthest_point_sampling(x, m)
    first = pointops.furthest_point_sampling(x, m0)
    assert (first == full[:, :m0]).all()
    cont = pointops.furthest_point_sampling(x, m, known=first)
    assert (cont == full).all()


@pytest.mark.parametrize("N,ncol,mode,span", [(20000, 4, 4, 40), (5000, 4, 3, 12), (3000, 3, 4, 9), (4000, 4, 1, 10),
                                              (4000, 4, 2, 10), (1, 4, 4, 3), (777, 4, 0, 2000)])
def test_voxelize_idx_gpu_bit_exact(hip, oracle, N, ncol, mode, span):
    """GPU voxelize_idx (row f2) vs the oracle's restatement of voxelize.cpp: voxel order, maps and rule rows."""
    from geoformer_amd import pointops

    rng = np.random.default_rng(N + mode)
    c = rng.integers(0, span, (N, ncol)).astype(np.int64)
    if ncol == 4:
        c[:, 0] = np.sort(rng.integers(0, 3, N))  # batch column, contiguous scenes
    if mode == 0:
        c = np.unique(c, axis=0)  # mode 0 = "guaranteed unique"
        c = c[rng.permutation(c.shape[0])]
    oc, p2v, v2p = oracle.voxelize_idx(c, mode)
    goc, gp2v, gv2p = pointops.voxelize_idx(torch.from_numpy(c).cuda(), mode)
    assert (gp2v.cpu().numpy() == p2v).all()
    assert gv2p.shape == v2p.shape and (gv2p.cpu().numpy() == v2p).all()
    assert (goc.cpu().numpy() == oc).all()


def test_voxelize_idx_gpu_rejects_out_of_range(hip):
    from geoformer_amd import _lib, pointops

    c = torch.tensor([[0, 1, 2, 3], [0, 70000, 2, 3]], dtype=torch.int64).cuda()
    with pytest.raises(_lib.GeoFormerHipError):
        pointops.voxelize_idx(c, 4)


@pytest.mark.parametrize("n,k,ahead", [(60_133, 50_000, 215_000), (30_000, 30_000, 0), (50_001, 50_000, 1000), (7, 7, 0),
                                       (120_000, 50_000, 215_000)])
def test_draw_sample_equals_numpy_draw_and_gather(hip, n, k, ahead):
    """pointops.draw_sample (gf_host_draw_sample: the per-scene draw of geoformer.py:575-579 as 32-bit indices into pinned
    memory + upload + gather, one native call) against numpy's own draw and an indexed copy: same indices, same points,
    same generator state afterwards -- with the generator's words drawn ahead (enough / too few) and without; twice in a row
    on the same buffers (the pinned buffer is rewritten in stream order)."""
    from geoformer_amd import pointops

    rng = np.random.default_rng(n)
    xyz = torch.from_numpy(rng.standard_normal((n + 100, 3)).astype(np.float32)).cuda()
    for rep in range(2):
        np.random.seed(5 + rep)
        np.random.rand(17 * rep)
        ref = np.random.choice(n, k, replace=False)
        ref_after = np.random.get_state()
        np.random.seed(5 + rep)
        np.random.rand(17 * rep)
        bufs = pointops.draw_sample_buffers(50_000, n, xyz.device)
        if ahead:
            pointops.legacy_prefetch(ahead)
        got = pointops.draw_sample(n, k, xyz, bufs)
        assert got is not None
        idx, pts = got
        got_after = np.random.get_state()
        assert idx.dtype == torch.int64 and tuple(idx.shape) == (k,) and tuple(pts.shape) == (1, k, 3) and pts.is_contiguous()
        assert (idx.cpu().numpy() == ref).all()
        assert torch.equal(pts[0], xyz[torch.from_numpy(ref).cuda()])
        assert (ref_after[1] == got_after[1]).all() and ref_after[2:] == got_after[2:]


def test_draw_sample_declines_what_it_cannot_do(hip):
    """More indices than the buffers hold, k > n: None (the caller takes legacy_choice's route, numpy's own errors)."""
    from geoformer_amd import pointops

    xyz = torch.zeros((100, 3), device="cuda")
    bufs = pointops.draw_sample_buffers(50, 100, xyz.device)
    st = np.random.get_state()
    assert pointops.draw_sample(100, 60, xyz, bufs) is None
    assert pointops.draw_sample(10, 11, xyz, bufs) is None
    assert (np.random.get_state()[1] == st[1]).all()
