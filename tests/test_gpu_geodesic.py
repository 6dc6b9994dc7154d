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


def _ms_forms(lib):
    """Both forms of the multi-source hop: the plain gather (default) and LDS tiles (GF_BFS_MS_TILES=1)."""
    lib.gf_dev_bfs_ms_tiles.argtypes = [__import__("ctypes").c_int]
    for tiles in (1, 0):
        lib.gf_dev_bfs_ms_tiles(tiles)
        yield tiles
    lib.gf_dev_bfs_ms_tiles(-1)


@pytest.mark.parametrize("n,nq,max_step", [(6000, 16, 256), (20000, 64, 128), (20000, 100, 40), (20000, 128, 5),
                                           (12000, 256, 256), (6000, 300, 64)])
def test_bfs_multi_source_matches_oracle(hip, oracle, n, nq, max_step):
    """gf_geodesic_bfs_ms (csrc/geodesic_ms.hip: all queries as bit lanes of cumulative reach masks, one launch per
    hop) against the oracle, bit for bit, for 1..5 mask words per vertex (nq = 300 takes the runtime-width gather
    kernel), hop limits that end the search early and late, duplicate sources, and both forms of the hop."""
    from geoformer_amd import pointops

    xyz = _pts(n, 5 + n)
    n = xyz.shape[0]  # (the generator returns about the number of points asked for)
    k, radius = 64, 0.05
    D, I = _ref_graph(oracle, xyz, k, radius)
    rng = np.random.default_rng(1)
    src = rng.integers(0, n, nq)
    src[-1] = src[0]  # two queries from one vertex
    ref = oracle.geodesic(D[:, 1:], I[:, 1:], src, radius, max_step)
    gd, gi, deg = pointops.knn_radius(_dev(xyz), k, radius)
    for tiles in _ms_forms(hip):
        geo = pointops.geodesic_bfs_ms(gd, gi, _dev(src.astype(np.int32)), radius, max_step).cpu().numpy()
        assert ((geo >= 0) == (ref >= 0)).all(), tiles
        assert (geo == ref).all(), tiles
    # an unsorted, unpadded table (every row the brute-force 64 nearest, beyond the radius too): rows need not be
    # distance-sorted or radius-limited for this kernel
    perm = rng.permutation(k - 1) + 1
    Dp = np.ascontiguousarray(D[:, np.r_[0, perm]].astype(np.float32))
    Ip = np.ascontiguousarray(I[:, np.r_[0, perm]].astype(np.int32))
    ref_p = oracle.geodesic(Dp[:, 1:], Ip[:, 1:].astype(np.int64), src, radius, max_step)
    geo = pointops.geodesic_bfs_ms(_dev(Dp), _dev(Ip), _dev(src.astype(np.int32)), radius, max_step).cpu().numpy()
    assert (geo == ref_p).all()


def test_bfs_multi_source_long_in_lists_and_duplicates(hip, oracle):
    """In-lists longer than the 16 fixed-width slots (a dense cube: every vertex has ~63 in-neighbours, the rest comes
    from the reverse CSR), 100 copies of one point, rows whose column 0 is not the vertex itself -- against the oracle,
    both forms of the hop; and the per-query kernel on the same inputs."""
    from geoformer_amd import pointops

    k, radius = 64, 0.05
    rng = np.random.default_rng(9)
    xyz = (rng.random((9000, 3)) * 0.09).astype(np.float32)
    D, I = _ref_graph(oracle, xyz, k, radius)
    inr = D <= np.float32(radius)
    Dm = np.where(inr, D, np.inf).astype(np.float32)
    Im = np.where(inr, I, -1).astype(np.int32)
    src = rng.integers(0, 9000, 70)
    ref = oracle.geodesic(D[:, 1:], I[:, 1:], src, radius, 64)
    for tiles in _ms_forms(hip):
        geo = pointops.geodesic_bfs_ms(_dev(Dm), _dev(Im), _dev(src.astype(np.int32)), radius, 64).cpu().numpy()
        assert (geo == ref).all(), tiles
    xyz = _pts(20000, 77)
    xyz[1000:1100] = xyz[5]
    D, I = _ref_graph(oracle, xyz, k, radius)
    src = rng.integers(0, xyz.shape[0], 48)
    src[:3] = [5, 1003, 1099]
    ref = oracle.geodesic(D[:, 1:], I[:, 1:], src, radius, 256)
    gd, gi, deg = pointops.knn_radius(_dev(xyz), k, radius)
    for tiles in _ms_forms(hip):
        geo = pointops.geodesic_bfs_ms(gd, gi, _dev(src.astype(np.int32)), radius, 256).cpu().numpy()
        assert (geo == ref).all(), tiles
    old = pointops.geodesic_bfs(gd, gi, deg, _dev(src.astype(np.int32)), radius, 256, wg_threads=512).cpu().numpy()
    assert (old == ref).all()


def test_bfs_multi_source_one_launch_and_batched_scenes(hip, oracle):
    """The two further forms of the multi-source search against the oracle: (i) ONE launch of resident tile workgroups
    that exchange their rows through memory (GF_BFS_MS_PERSIST; spatial working order from the coordinates; its bounded
    waits must not time out), (ii) several scenes searched together (gf_geodesic_bfs_ms_sets: one hop launch serves all
    scenes of a batch)."""
    from geoformer_amd import pointops

    hip.gf_dev_bfs_ms_persist.argtypes = [__import__("ctypes").c_int]
    k, radius = 64, 0.05
    graphs, srcs, refs, xyzs = [], [], [], []
    for n, nq_seed in ((20000, 3), (9000, 4), (14000, 5)):
        xyz = _pts(n, 31 + n)
        n = xyz.shape[0]
        D, I = _ref_graph(oracle, xyz, k, radius)
        src = np.random.default_rng(nq_seed).integers(0, n, 128)
        refs.append(oracle.geodesic(D[:, 1:], I[:, 1:], src, radius, 96))
        gd, gi, deg = pointops.knn_radius(_dev(xyz), k, radius)
        graphs.append((gd, gi)); srcs.append(_dev(src.astype(np.int32))); xyzs.append(_dev(xyz))
    hip.gf_dev_bfs_ms_persist(1)
    try:
        for (gd, gi), src, xyz, ref in zip(graphs, srcs, xyzs, refs):
            geo, flag = pointops.geodesic_bfs_ms(gd, gi, src, radius, 96, xyz=xyz, return_flag=True)
            assert int(flag.item()) == 0, "a wait of the one-launch search timed out"
            assert (geo.cpu().numpy() == ref).all()
    finally:
        hip.gf_dev_bfs_ms_persist(-1)
    # the Morton working order with one launch per hop (tiles)
    geo = pointops.geodesic_bfs_ms(graphs[0][0], graphs[0][1], srcs[0], radius, 96, xyz=xyzs[0])
    assert (geo.cpu().numpy() == refs[0]).all()
    geos = pointops.geodesic_bfs_ms_batch(graphs, srcs, radius, 96)
    for g, ref in zip(geos, refs):
        assert (g.cpu().numpy() == ref).all()


def test_gated_sampling_and_search_equal_the_two_launch_form(hip):
    """One sampling launch with a gate + the search launched BESIDE it on a second stream (it waits for the first nq picks
    inside the kernel) == sampling, then search.  With and without the LDS pad that keeps the search off the sampler's
    compute units; the time-out word stays clear."""
    import numpy as np
    import torch

    from geoformer_amd import pointops, scene

    p = scene.make_scene(60_000, 77)["xyz"]
    xyz = torch.from_numpy(np.ascontiguousarray(p[:24_000])).cuda()
    n = xyz.shape[0]
    gd, gi, deg = pointops.knn_radius(xyz, 64, 0.05)
    pts = xyz[None].contiguous()
    nq, m = 96, 512
    ref_idx = pointops.furthest_point_sampling(pts, m)
    ref_geo = pointops.geodesic_bfs(gd, gi, deg, ref_idx[0, :nq].contiguous(), 0.05, 64, wg_threads=512)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for pad in (0, 88 * 1024):
        for wg in (512, 1024):
            idx, gate, reset_ev = pointops.furthest_point_sampling_gated(pts, m, nq, lds_pad=pad)
            side.wait_event(reset_ev)
            with torch.cuda.stream(side):
                geo = pointops.geodesic_bfs_gated(gd, gi, idx[0, :nq], 0.05, 64, gate, nq, wg_threads=wg)
            torch.cuda.synchronize()
            assert torch.equal(idx, ref_idx)
            assert int(gate[0].item()) >= nq and int(gate[2].item()) == 0
            assert torch.equal(geo, ref_geo), (pad, wg)
        # the forward's gated shape: buffers prepared ahead, 768 threads per query, LDS capped so that two share a unit
        prep = pointops.fps_gated_prepare(pts.device, m)
        idx, gate, reset_ev = pointops.furthest_point_sampling_gated(pts, m, nq, lds_pad=pad, prepared=prep)
        side.wait_event(reset_ev)
        with torch.cuda.stream(side):
            geo = pointops.geodesic_bfs_gated(gd, gi, idx[0, :nq], 0.05, 64, gate, nq, wg_threads=768, lds_cap=64 * 1024)
        torch.cuda.synchronize()
        assert torch.equal(idx, ref_idx) and int(gate[2].item()) == 0 and torch.equal(geo, ref_geo), pad
    # a gate beyond the last pick opens at the end of the launch
    idx, gate, reset_ev = pointops.furthest_point_sampling_gated(pts, 64, 1000)
    torch.cuda.synchronize()
    assert torch.equal(idx, ref_idx[:, :64]) and int(gate[0].item()) == 1000
