"""GPU parity at BASELINE config 2's real size: the S150k benchmark scene (150 269 points, ~142k voxels).

The launch shapes gf_conv_fwd picks depend on the level's size; the small-scene tests never reach the ones the
benchmark spends its time in.  Here the level-1 geometry of the benchmark scene itself goes through every
convolution entry point against the oracle (oracle/gf_oracle.c, scalar C: ~0.5 GFLOP per case = seconds), every
launch shape is forced on a small input through the dev knobs (include/geoformer_hip_dev.h), the BFS runs at the
benchmark's foreground size, at sizes that promote the workgroup 256 -> 512 -> 1024 and on tables that take the
global-memory kernel (K % 4 != 0, n > 2^19), and one whole eval forward of the scene is compared stage by stage
with the same forward on the host through the oracle's operators.
Tolerances: integers bit-exact, floats <= 1e-4 abs (BASELINE.json north_star); the whole-forward comparison allows
32 fp32 ulps of a tensor's largest magnitude where that exceeds 1e-4 (see _close)."""
import numpy as np
import pytest
import torch

from tests.util import random_voxels

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _close(got, ref, ulps=32):
    """1e-4 abs (BASELINE.json north_star), or `ulps` fp32 epsilons of the tensor's largest magnitude where that is
    more: 71 convolutions deep the activations of the randomly initialised U-Net reach |x| ~ 60-70 and two fp32
    evaluations with different summation orders drift apart by ~1e-4 there (tools/parity_by_stage.py: the
    difference doubles level by level, mean 1e-6, no single stage stands out).  ONE bound, stated in DESIGN.md 2 and
    used by bench.py as well: an fp32 path against the float64 arbiter 32 epsilons; the DIFFERENCE of two fp32 paths
    (GPU against the oracle-backed host forward) 64 = the sum of two such errors (`_close_2fp32`)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    tol = max(1e-4, ulps * 1.1920929e-07 * float(np.abs(ref).max()))
    return float(np.abs(got - ref).max()) < tol


def _close_2fp32(got, ref):
    return _close(got, ref, ulps=64)


@pytest.fixture(scope="module")
def s150k():
    from geoformer_amd import scene

    sc = scene.make_scene(150_000, 1234)
    batch = scene.make_batch([sc])
    coords = batch["voxel_locs"].numpy().astype(np.int32)
    shape = tuple(int(s) for s in batch["spatial_shape"])
    return sc, batch, coords, shape


@pytest.fixture(scope="module")
def level1(hip, oracle, s150k):
    from geoformer_amd import sparse

    _, _, coords, shape = s150k
    nbr = oracle.rules_subm3(coords, shape)
    c = _dev(coords)
    rules = sparse.subm_rules(c, sparse.build_index(c, 1, shape))
    assert coords.shape[0] >= 110_000 and (coords.shape[0] + 15) // 16 >= 6900
    assert (rules.nbr.cpu().numpy() == nbr).all()  # rulebook bit-exact at full size
    return coords.shape[0], nbr, rules


@pytest.mark.parametrize("Cin,Cout", [(6, 16), (16, 16), (32, 16), (16, 32)])
def test_conv_fwd_full_size(oracle, level1, Cin, Cout):
    """plain / BN+ReLU prologue / residual epilogue / both, on the level-1 table of the benchmark scene."""
    from geoformer_amd import sparse

    M, nbr, rules = level1
    rng = np.random.default_rng(Cin * 17 + Cout)
    feats = rng.standard_normal((M, Cin)).astype(np.float32)
    W = (rng.standard_normal((27, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cin).astype(np.float32)
    shift = (0.3 * rng.standard_normal(Cin)).astype(np.float32)
    res = rng.standard_normal((M, Cout)).astype(np.float32)
    act = np.maximum(feats * scale + shift, 0).astype(np.float32)
    ref_plain = oracle.conv_fwd(feats, W, nbr, M)
    ref_act = oracle.conv_fwd(act, W, nbr, M)
    x, w = _dev(feats), _dev(W)
    kw = dict(in_scale=_dev(scale), in_shift=_dev(shift))
    got = sparse.conv_fwd(x, w, rules.nbr, rules.gmask, 27, M, rules.ld).cpu().numpy()
    assert np.abs(got - ref_plain).max() < 1e-4
    got = sparse.conv_fwd(x, w, rules.nbr, rules.gmask, 27, M, rules.ld, **kw).cpu().numpy()
    assert np.abs(got - ref_act).max() < 1e-4
    got = sparse.conv_fwd(x, w, rules.nbr, rules.gmask, 27, M, rules.ld, residual=_dev(res)).cpu().numpy()
    assert np.abs(got - (ref_plain + res)).max() < 1e-4
    got = sparse.conv_fwd(x, w, rules.nbr, rules.gmask, 27, M, rules.ld, residual=_dev(res), **kw).cpu().numpy()
    assert np.abs(got - (ref_act + res)).max() < 1e-4
    # the same four through the step table (the counted-loop kernels where the shape allows them), plus the
    # epilogue activation max(out * s + t, 0)
    osc = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    osh = (0.3 * rng.standard_normal(Cout)).astype(np.float32)
    st = dict(steps=rules.steps)
    got = sparse.conv_fwd(x, w, rules.nbr, rules.gmask, 27, M, rules.ld, **st).cpu().numpy()
    assert np.abs(got - ref_plain).max() < 1e-4
    got = sparse.conv_fwd(x, w, rules.nbr, rules.gmask, 27, M, rules.ld, **kw, **st).cpu().numpy()
    assert np.abs(got - ref_act).max() < 1e-4
    got = sparse.conv_fwd(x, w, rules.nbr, rules.gmask, 27, M, rules.ld, residual=_dev(res), **st).cpu().numpy()
    assert np.abs(got - (ref_plain + res)).max() < 1e-4
    got = sparse.conv_fwd(x, w, rules.nbr, rules.gmask, 27, M, rules.ld, residual=_dev(res), **kw, **st).cpu().numpy()
    assert np.abs(got - (ref_act + res)).max() < 1e-4
    for use_steps in (st, {}):
        got = sparse.conv_fwd(x, w, rules.nbr, rules.gmask, 27, M, rules.ld, out_scale=_dev(osc), out_shift=_dev(osh),
                              **kw, **use_steps).cpu().numpy()
        assert np.abs(got - np.maximum(ref_act * osc + osh, 0)).max() < 1e-4
    # the 1x1x1 identity-branch conv (K == 1, no table) at the same size
    W1 = (rng.standard_normal((1, Cin, Cout)) / np.sqrt(Cin)).astype(np.float32)
    got = sparse.conv_fwd(x, _dev(W1), None, None, 1, M, 0).cpu().numpy()
    assert np.abs(got - feats @ W1[0]).max() < 1e-4


@pytest.mark.parametrize("Cin,Cout", [(16, 16), (32, 16)])
def test_resblock_full_size(oracle, level1, Cin, Cout):
    """gf_resblock_fwd (two 3x3x3 launches + the 1x1x1 identity conv when the widths differ) at level-1 size."""
    from geoformer_amd import sparse

    M, nbr, rules = level1
    rng = np.random.default_rng(Cin + 1000)
    x = rng.standard_normal((M, Cin)).astype(np.float32)
    W0 = (rng.standard_normal((27, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    W1 = (rng.standard_normal((27, Cout, Cout)) / np.sqrt(9 * Cout)).astype(np.float32)
    Wi = (rng.standard_normal((1, Cin, Cout)) / np.sqrt(Cin)).astype(np.float32) if Cin != Cout else None
    s0, s1 = (rng.uniform(0.5, 1.5, c).astype(np.float32) for c in (Cin, Cout))
    t0, t1 = ((0.3 * rng.standard_normal(c)).astype(np.float32) for c in (Cin, Cout))
    h = oracle.conv_fwd(np.maximum(x * s0 + t0, 0).astype(np.float32), W0, nbr, M)
    ref = oracle.conv_fwd(np.maximum(h * s1 + t1, 0).astype(np.float32), W1, nbr, M) + (x if Wi is None else x @ Wi[0])
    wp = [sparse.pack_weights(_dev(w)) for w in (W0, W1)]
    wpi = None if Wi is None else sparse.pack_weights(_dev(Wi))
    got = sparse.resblock_fwd(_dev(x), wp[0], wp[1], wpi, rules.nbr, rules.gmask, 27, M, rules.ld, Cin, Cout,
                              _dev(s0), _dev(t0), _dev(s1), _dev(t1)).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-4


@pytest.mark.parametrize("Cin,Cout", [(6, 16), (16, 16), (32, 16), (16, 32)])
def test_conv_dgrad_wgrad_full_size(oracle, level1, Cin, Cout):
    from geoformer_amd import sparse

    M, nbr, rules = level1
    rng = np.random.default_rng(Cin * 5 + Cout * 3)
    feats = rng.standard_normal((M, Cin)).astype(np.float32)
    gout = rng.standard_normal((M, Cout)).astype(np.float32)
    W = (rng.standard_normal((27, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    ref_d = oracle.conv_dgrad(gout, W, nbr, M)
    got_d = sparse.conv_dgrad(_dev(gout), _dev(W), ("subm", (rules.nbr, rules.gmask, 27, M, rules.ld)), M).cpu().numpy()
    assert np.abs(got_d - ref_d).max() < 1e-4 * max(1.0, float(np.abs(ref_d).max()))
    ref_w = oracle.conv_wgrad(feats, gout, nbr, 27)
    got_w = sparse.conv_wgrad(_dev(feats), _dev(gout), rules.nbr, 27, M, rules.ld).cpu().numpy()
    # sums of ~35 000 products of unit-variance terms (|dW| ~ 200): relative to the magnitude
    assert np.abs(got_w - ref_w).max() < 2e-4 * max(1.0, float(np.abs(ref_w).max()))
    # the route the training step takes: (group, offset) pairs without a neighbour skipped by the table's group masks
    got_m = sparse.conv_wgrad(_dev(feats), _dev(gout), rules.nbr, 27, M, rules.ld, gmask=rules.gmask).cpu().numpy()
    assert np.abs(got_m - ref_w).max() < 2e-4 * max(1.0, float(np.abs(ref_w).max()))


def test_down_up_full_size(oracle, s150k, hip):
    """Strided 2x2x2 conv and its inverse between levels 1 and 2 of the benchmark scene (54k coarse voxels)."""
    from geoformer_amd import sparse

    _, _, coords, shape = s150k
    M = coords.shape[0]
    rng = np.random.default_rng(77)
    feats = rng.standard_normal((M, 16)).astype(np.float32)
    Wd = (rng.standard_normal((8, 16, 32)) / np.sqrt(3 * 16)).astype(np.float32)
    Wu = (rng.standard_normal((8, 32, 16)) / np.sqrt(32)).astype(np.float32)
    oc, child, parent, koff = oracle.rules_down2(coords, shape)
    r = sparse.down_rules(_dev(coords), 1, shape)
    assert r.M_out == oc.shape[0] and (r.out_coords.cpu().numpy() == oc).all()
    assert (r.child.cpu().numpy()[:, : child.shape[1]] == child).all()
    ref_d = oracle.conv_fwd(feats, Wd, child, oc.shape[0])
    ref_u = oracle.conv_fwd(ref_d, Wu, oracle.up_table(parent, koff), M)
    d = sparse.conv_fwd(_dev(feats), _dev(Wd), r.child, r.gmask_down, 8, r.M_out, r.ld)
    u = sparse.conv_fwd(d, _dev(Wu), r.up, r.gmask_up, 8, M, r.ld_up)
    assert np.abs(d.cpu().numpy() - ref_d).max() < 1e-4
    assert np.abs(u.cpu().numpy() - ref_u).max() < 1e-4
    # level 2 of the scene (C = 32, ~3400 groups: the split launch shape at its real size)
    nbr2 = oracle.rules_subm3(oc, r.out_shape)
    rules2 = sparse.subm_rules(r.out_coords.contiguous(), r.index_out)
    assert (rules2.nbr.cpu().numpy() == nbr2).all()
    f2 = rng.standard_normal((oc.shape[0], 32)).astype(np.float32)
    W2 = (rng.standard_normal((27, 32, 32)) / np.sqrt(9 * 32)).astype(np.float32)
    got = sparse.conv_fwd(_dev(f2), _dev(W2), rules2.nbr, rules2.gmask, 27, oc.shape[0], rules2.ld).cpu().numpy()
    assert np.abs(got - oracle.conv_fwd(f2, W2, nbr2, oc.shape[0])).max() < 1e-4


KNOBS = [dict(split=0, pair=1, g16=0), dict(split=0, pair=0, g16=0),
         dict(split=1, wide=0, g16=0), dict(split=1, wide=1, g16=0), dict(split=0, pair=0, block=64, g16=0),
         dict(split=0, pair=0, block=128, g16=0),
         # the counted-loop kernels over the step table (16 output channels, Cin 16 / 32 only; other widths fall
         # through to the size-based choice): plain, LDS weights, several groups per wave, pipelined over chunks
         dict(g16=1, g16_ldsw=0, g16_pipe=0), dict(g16=1, g16_ldsw=1, g16_pipe=0, g16_gpw=3),
         dict(g16=1, g16_ldsw=0, g16_pipe=1), dict(g16=1, g16_ldsw=1, g16_pipe=1)]
# none of the above may fall into the flat-chain kernel of the deep levels (size-based for small inputs); it is a shape of its own
KNOBS = [dict(k, flat=0) for k in KNOBS] + [dict(flat=1)]


@pytest.mark.parametrize("Cin,Cout", [(16, 16), (32, 16), (6, 16), (32, 32), (48, 64), (19, 21)])
def test_every_launch_shape_forced(hip, oracle, Cin, Cout):
    """gf_conv_fwd chooses a launch shape from the level's size; each shape is forced here on one small input
    (dev knobs, include/geoformer_hip_dev.h) with prologue and residual, and must give the oracle's result."""
    from geoformer_amd import sparse

    rng = np.random.default_rng(Cin * 7 + Cout)
    shape = (40, 36, 30)
    coords = random_voxels(rng, 3000, shape, 1, surface=True)
    M = coords.shape[0]
    feats = rng.standard_normal((M, Cin)).astype(np.float32)
    W = (rng.standard_normal((27, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cin).astype(np.float32)
    shift = (0.3 * rng.standard_normal(Cin)).astype(np.float32)
    res = rng.standard_normal((M, Cout)).astype(np.float32)
    nbr = oracle.rules_subm3(coords, shape)
    ref = oracle.conv_fwd(np.maximum(feats * scale + shift, 0).astype(np.float32), W, nbr, M) + res
    ref_plain = oracle.conv_fwd(feats, W, nbr, M)
    c = _dev(coords)
    old_min = sparse.STEPS_MIN_ROWS
    sparse.STEPS_MIN_ROWS = 0  # build the step table for this small voxel set too
    try:
        rules = sparse.subm_rules(c, sparse.build_index(c, 1, shape))
    finally:
        sparse.STEPS_MIN_ROWS = old_min
    # the step table against the neighbour table it restates: present offsets in ascending order, four per entry
    st = rules.steps.cpu().numpy()
    ng = rules.ld // 16
    blocks = st[: ng * 7 * 64].reshape(ng, 7, 16, 4)
    for g in rng.integers(0, ng, 40):
        ks = [k for k in range(27) if (nbr[k, g * 16:(g + 1) * 16] >= 0).any()]
        for s_, k in enumerate(ks):
            assert (blocks[g, s_ // 4, :, s_ % 4] == nbr[k, g * 16:(g + 1) * 16]).all()
        for s_ in range(len(ks), max(12, (len(ks) + 3) // 4 * 4)):
            assert (blocks[g, s_ // 4, :, s_ % 4] == -1).all()
    tail = st[ng * 7 * 64:]
    nchunks = int(tail[0])
    bounds = tail[1: nchunks + 2]
    assert bounds[0] == 0 and bounds[-1] == (M + 15) // 16 and (np.diff(bounds) >= 0).all()
    x, w, s, t, r = _dev(feats), _dev(W), _dev(scale), _dev(shift), _dev(res)
    try:
        for knobs in KNOBS:
            sparse.dev_conv_knobs(**knobs)
            got = sparse.conv_fwd(x, w, rules.nbr, rules.gmask, 27, M, rules.ld, in_scale=s, in_shift=t, residual=r,
                                  steps=rules.steps)
            assert np.abs(got.cpu().numpy() - ref).max() < 1e-4, knobs
            got = sparse.conv_fwd(x, w, rules.nbr, rules.gmask, 27, M, rules.ld, steps=rules.steps)
            assert np.abs(got.cpu().numpy() - ref_plain).max() < 1e-4, knobs
    finally:
        sparse.dev_conv_knobs()  # back to the size-based choice


# ---- geodesic BFS at and beyond the benchmark's size -----------------------------------------------------------
def _graph(xyz, k=64, radius=0.05):
    from geoformer_amd import pointops

    gd, gi, deg = pointops.knn_radius(_dev(xyz), k, radius, sqrt_out=True, check_overflow=True)
    return gd, gi, deg


def _scene_points(n, seed):
    from geoformer_amd import scene

    p = scene.make_scene(n, seed)["xyz"]
    return np.ascontiguousarray(p[np.random.default_rng(seed).permutation(p.shape[0])])


@pytest.mark.parametrize("n,nq,wg", [(60_108, 24, 256), (60_108, 8, 1024), (140_000, 12, 256), (300_000, 6, 256),
                                     (300_000, 4, 512)])
def test_bfs_large_scenes(hip, oracle, n, nq, wg):
    """n = 60 108 is the benchmark's foreground size at four queries per compute unit; 140 000 and 300 000 no longer
    fit the 256-thread (resp. 512-thread) share of LDS and are promoted to the next larger workgroup
    (gf_geodesic_bfs_cfg).  The graph comes from gf_knn_radius (itself checked against brute force in
    test_gpu_geodesic.py); both sides walk the same table."""
    from geoformer_amd import pointops

    xyz = _scene_points(n, 31 + n)[:n]
    n = xyz.shape[0]
    gd, gi, deg = _graph(xyz)
    D, I = gd.cpu().numpy(), gi.cpu().numpy()
    src = np.random.default_rng(n).integers(0, n, nq)
    ref = oracle.geodesic(D[:, 1:], I[:, 1:], src, 0.05, 256)
    geo = pointops.geodesic_bfs(gd, gi, deg, _dev(src.astype(np.int32)), 0.05, 256, wg_threads=wg).cpu().numpy()
    assert ((geo >= 0) == (ref >= 0)).all() and (geo == ref).all()
    assert (geo >= 0).sum(1).max() > n // 20  # the walk really spreads (not a trivially empty frontier)


@pytest.mark.parametrize("wg,qcap", [(512, 0), (1024, 0), (512, 300), (1024, 64)])
def test_bfs_kernels_queue_overflow_and_long_rings(hip, oracle, wg, qcap):
    """The LDS-resident search (k_geodesic_bfs_lds) on a 90 000-point foreground against the oracle, bit for bit, with the
    LDS queue capacity as the launch derives it and cut to 300 / 64 entries (gf_dev_bfs_qcap_max): rings of up to ~2 000
    vertices then live mostly in the global overflow."""
    from geoformer_amd import _lib, pointops

    lib = _lib.load()
    n = 90_000
    xyz = _scene_points(n, 977)[:n]
    gd, gi, deg = _graph(xyz)
    D, I = gd.cpu().numpy(), gi.cpu().numpy()
    src = np.random.default_rng(3).integers(0, n, 10)
    ref = oracle.geodesic(D[:, 1:], I[:, 1:], src, 0.05, 256)
    lib.gf_dev_bfs_qcap_max(qcap)
    try:
        geo = pointops.geodesic_bfs(gd, gi, deg, _dev(src.astype(np.int32)), 0.05, 256, wg_threads=wg).cpu().numpy()
        short = pointops.geodesic_bfs(gd, gi, deg, _dev(src.astype(np.int32)), 0.05, 7, wg_threads=wg).cpu().numpy()
    finally:
        lib.gf_dev_bfs_qcap_max(0)
    assert (geo == ref).all()
    assert (short == oracle.geodesic(D[:, 1:], I[:, 1:], src, 0.05, 7)).all()  # the max_step cut with levels in flight
    reached = (geo >= 0).sum(1)
    assert reached.max() > n // 3


@pytest.mark.parametrize("case", ["K63", "n>2^19"])
def test_bfs_global_memory_kernel(hip, oracle, case):
    """Tables the LDS-resident kernel does not take: a column count that is not a multiple of four, and more
    points than its bitmaps hold (k_geodesic_bfs: visited state and queues in global memory)."""
    from geoformer_amd import pointops

    if case == "K63":
        from geoformer_amd import scene

        n = 30_000  # a small room at ScanNet density (24 mm spacing), so the 5 cm graph is connected
        xyz = np.ascontiguousarray(scene.make_scene(n, 3, room=(2.5, 2.0, 1.2), n_boxes=2)["xyz"][:n])
        gd, gi, deg = _graph(xyz)
        gd, gi = gd[:, :63].contiguous(), gi[:, :63].contiguous()
        deg = torch.clamp(deg, max=62)
    else:
        n = (1 << 19) + 4097
        xyz = _scene_points(n + 5000, 9)[:n]
        gd, gi, deg = _graph(xyz)
    n = xyz.shape[0]
    D, I = gd.cpu().numpy(), gi.cpu().numpy()
    src = np.random.default_rng(5).integers(0, n, 5)
    ref = oracle.geodesic(D[:, 1:], I[:, 1:], src, 0.05, 200)
    geo = pointops.geodesic_bfs(gd, gi, deg, _dev(src.astype(np.int32)), 0.05, 200).cpu().numpy()
    assert (geo == ref).all()
    assert (geo >= 0).sum(1).max() > 1000


# ---- the whole forward of the benchmark scene ---------------------------------------------------------------------
def _forward_gpu_against_host(s150k, close, state=None):
    """One eval forward of the S150k benchmark scene on the GPU against the same forward of the build's model on the
    host through the oracle's scalar operators (oracle/cpu_backend.py; what bench.py times as cpu_baseline).
    The class decision of 150k points under random weights has a handful of near-ties, so the host run takes the
    foreground decision from its OWN backbone only after checking it against the GPU's to 1e-4, and then continues
    from the GPU's backbone output: everything downstream (host draw, FPS picks, kNN rows, BFS, decoder, mask head)
    is compared on identical point sets."""
    from bench import build_model, to_device
    from oracle import cpu_backend

    _, batch, _, _ = s150k
    dev_batch = to_device(batch, "cuda")
    if state is None:
        m = build_model("cuda", probe_batch=dev_batch)
    else:  # the same weights on both sides, background shift included
        m = build_model("cuda")
        m.load_state_dict(state)
    shift = m._bench_bias_shift
    cap = {}

    def wrap(model, tag):
        dec = model.forward_decoder

        def dec_w(cl, cf, ql, pc, geo, pei):
            r = dec(cl, cf, ql, pc, geo, pei)
            cap[tag] = dict(context_locs=cl.detach().cpu().numpy(), context_feats=cf.detach().cpu().numpy(),
                            pre_enc_inds=pei.detach().cpu().numpy(), geo=geo[0].detach().cpu().numpy(),
                            dec=r.detach().cpu().numpy())
            return r

        model.forward_decoder = dec_w

    wrap(m, "gpu")
    np.random.seed(11)
    with torch.no_grad():
        out = m(dev_batch, 300, training=False)
    torch.cuda.synchronize()
    g_sem = out["semantic_scores"].cpu().numpy()
    g_fg = out["fg_idxs"].cpu().numpy()
    # the GPU model's per-point backbone features (the fused path does not materialise them: gathered here)
    with torch.no_grad():
        feats_g = m.forward_backbone(dev_batch, 1, want_preds=False)[0]
        out_feats_gpu = (feats_g[0][feats_g[1].long()] if isinstance(feats_g, tuple) else feats_g).cpu()

    with cpu_backend.installed(), torch.no_grad():
        mc = build_model("cpu", bias_shift=shift)
        if state is not None:
            mc.load_state_dict(state)
        wrap(mc, "cpu")
        fb = mc.forward_backbone
        stats = {}

        def fb_w(batch_input, batch_size, want_preds=True):
            feats, sem, preds = fb(batch_input, batch_size, want_preds=True)
            c_sem = sem.numpy()
            stats["sem_maxabs"] = float(np.abs(c_sem - g_sem).max())
            assert stats["sem_maxabs"] < 1e-4, stats
            c_fg = np.nonzero(c_sem.argmax(1) >= 4)[0]
            diff = np.setxor1d(c_fg, g_fg)
            stats["fg_diff"] = int(diff.size)
            # points on which the two class decisions differ are ties at the tolerance
            top2 = np.sort(c_sem[diff], axis=1)[:, -2:] if diff.size else np.zeros((0, 2))
            assert diff.size <= 8 and (top2[:, 1] - top2[:, 0] < 2e-4).all()
            # continue from the GPU's backbone output (within 1e-4 of this one): identical foreground sets
            g_feats = out_feats_gpu
            assert close(g_feats.numpy(), feats.numpy())
            sem_g = torch.from_numpy(g_sem)
            return g_feats, sem_g, sem_g.max(1)[1]

        mc.forward_backbone = fb_w
        np.random.seed(11)
        outc = mc(batch, 300, training=False)

    assert (outc["fg_idxs"].numpy() == g_fg).all()
    g, c = cap["gpu"], cap["cpu"]
    assert (m.last_sampling_indices.cpu().numpy() == mc.last_sampling_indices.numpy()).all()  # host draw
    assert (g["pre_enc_inds"] == c["pre_enc_inds"]).all()  # 2048 FPS picks among 50 000 points, bit-exact
    assert (g["context_locs"] == c["context_locs"]).all()
    assert close(g["context_feats"], c["context_feats"])
    assert ((g["geo"] >= 0) == (c["geo"] >= 0)).all() and (g["geo"] == c["geo"]).all()  # reach sets and fp32 sums
    assert close(g["dec"][-1], c["dec"][-1])
    mpg, mpc = out["mask_predictions"][-1], outc["mask_predictions"][-1]
    assert close(mpg["cls_logits"].cpu().numpy(), mpc["cls_logits"].numpy())
    mlg, mlc = mpg["mask_logits"][0].cpu().numpy(), mpc["mask_logits"][0].numpy()
    assert mlg.shape == mlc.shape == (256, g_fg.shape[0])
    assert close(mlg, mlc)
    pg, pc = out["proposal_scores"], outc["proposal_scores"]
    assert len(pg[0]) == len(pc[0])
    if len(pg[0]):
        assert (pg[0].cpu().numpy() == pc[0].numpy()).all()
        assert np.abs(pg[1].cpu().numpy() - pc[1].numpy()).max() < 1e-4
        d = np.abs(pg[2].sum(1).cpu().numpy() - pc[2].sum(1).numpy())
        assert d.max() <= 3
    return {"mask_logits_scale": float(np.abs(mlc).max()), "mask_logits_err": float(np.abs(mlg - mlc).max()),
            "sem_err": stats["sem_maxabs"], "dec_err": float(np.abs(g["dec"][-1] - c["dec"][-1]).max()),
            "context_feats_scale": float(np.abs(c["context_feats"]).max()), "n_fg": int(g_fg.shape[0])}


def test_forward_s150k_matches_oracle_backed_forward(hip, oracle, s150k):
    """The random-init benchmark model (activations up to |x| ~ 60): two fp32 evaluations, so every stage to max(1e-4,
    64 fp32 epsilons of the tensor's largest magnitude) -- `_close_2fp32`, the bound bench.py's parity_s150k states; the
    float64 arbiter below holds EACH fp32 path to 32 epsilons of double (measured there: GPU 20, host 25 on the mask
    logits, 31-32 between the two here)."""
    _forward_gpu_against_host(s150k, _close_2fp32)


def test_forward_s150k_calibrated_weights_hold_1e4_absolute(hip, oracle, s150k):
    """north_star's bound LITERALLY: the same comparison with BatchNorm statistics matched to the scene's own activations
    (tests/util.calibrated_benchmark_state: what a trained network has), every float stage to 1e-4 ABSOLUTE, no
    magnitude-relative allowance -- semantic scores, context features, decoder output, class logits, mask logits,
    proposal scores -- and all integer stages bit-exact as before."""
    from tests.util import calibrated_benchmark_state

    _, batch, _, _ = s150k
    state, _ = calibrated_benchmark_state(batch)

    seen = []

    def close_abs(got, ref):
        d = float(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).max())
        seen.append(d)
        assert d < 1e-4, ("stage maxima so far", seen)
        return True

    info = _forward_gpu_against_host(s150k, close_abs, state=state)
    print("calibrated S150k forward:", info)
    assert info["n_fg"] > 30_000  # (the scene keeps a real foreground under the calibrated weights)


def test_conv_probe_events_s150k(hip, s150k):
    """bench.py's roofline probe on the benchmark scene: the seven level-1 16->16 launches of a forward are recorded in
    both modes (structure only: how long a launch took on this box is no business of a parity test), and binding
    events to the launches does not change what the forward computes."""
    import bench
    from geoformer_amd import _lib

    _, batch, _, _ = s150k
    dev_batch = bench.to_device(batch, "cuda")
    m = bench.build_model("cuda", probe_batch=dev_batch)
    lib = _lib.load()
    outs = {}
    for mode in (0, 1, 3):
        lib.gf_dev_unet_probe(mode)
        try:
            np.random.seed(5)
            with torch.no_grad():
                outs[mode] = m(dev_batch, 0, training=False)["semantic_scores"].clone()
            torch.cuda.synchronize()
        finally:
            lib.gf_dev_unet_probe(0)
        recs = bench._read_probe()
        if mode == 0:
            assert recs == []
            continue
        assert len(recs) == 7 and all(r[3] == 16 and r[4] == 16 and r[2] == 27 for r in recs)
        assert all(np.isfinite(r[9]) for r in recs)
        if mode == 1:
            assert all(r[10] == -1.0 for r in recs)
        else:
            assert all(np.isfinite(r[10]) and r[10] != -1.0 for r in recs), recs
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[3])


# ---- float64 arbiter: GPU fp32 and host fp32 against the same forward in double precision --------------------------
ARBITER_EPS = 32  # fp32 epsilons (2^-23) of a tensor's largest magnitude; DESIGN.md section 2 states the bound


def _arb_close(got, ref64, what, log):
    got, ref64 = np.asarray(got, np.float64), np.asarray(ref64, np.float64)
    scale = float(np.abs(ref64).max())
    err = float(np.abs(got - ref64).max())
    tol = max(1e-4, ARBITER_EPS * 1.1920929e-07 * scale)
    log.append((what, err, scale, err / (1.1920929e-07 * max(scale, 1e-30)), tol))
    return err <= tol


def test_forward_s150k_fp32_paths_against_float64_arbiter(hip, oracle, s150k):
    """North-star parity is "<= 1e-4 abs"; 71 convolutions deep the random-init activations reach |x| ~ 60, where 1e-4
    is 14 fp32 epsilons and two fp32 evaluations with different summation orders cannot agree to it.  Instead of
    comparing the two fp32 paths with each other (and calling the difference rounding), BOTH are compared with an
    arbiter: the same forward of the build's model in double precision (oracle.cpu_backend f64 mode: float64
    convolutions / voxel means / gathers / torch modules; every integer decision -- FPS, ball query, kNN, BFS -- on the
    fp32 coordinates as in the fp32 runs).  No backbone re-injection: each run carries its own features end to end.
    Only the CLASS DECISION (an integer vector) of the GPU run is handed to the two host runs, after checking that
    their own decisions differ from it on near-ties only.  Bound for every float stage, GPU and host alike:
    max(1e-4, ARBITER_EPS * 2^-23 * max|arbiter|)."""
    from bench import build_model, to_device
    from oracle import cpu_backend
    from oracle import oracle as orc

    L = orc.lib()
    L.orc_set_threads.restype = int
    L.orc_set_threads(64)
    _, batch, _, _ = s150k
    dev_batch = to_device(batch, "cuda")
    m = build_model("cuda", probe_batch=dev_batch)
    shift = m._bench_bias_shift
    cap = {}

    def wrap(model, tag):
        dec = model.forward_decoder

        def dec_w(cl, cf, ql, pc, geo, pei):
            r = dec(cl, cf, ql, pc, geo, pei)
            cap[tag] = dict(context_locs=cl.detach().cpu().numpy(), context_feats=cf.detach().cpu().numpy(),
                            pre_enc_inds=pei.detach().cpu().numpy(), geo=geo[0].detach().cpu().numpy(),
                            dec=r.detach().cpu().numpy())
            return r

        model.forward_decoder = dec_w

    wrap(m, "gpu")
    np.random.seed(11)
    with torch.no_grad():
        out = m(dev_batch, 300, training=False)
    torch.cuda.synchronize()
    g_sem = out["semantic_scores"].cpu().numpy()
    g_preds = torch.from_numpy(g_sem).max(1)[1]
    g_fg = out["fg_idxs"].cpu().numpy()
    flips = {}

    def host(tag, f64):
        with cpu_backend.installed(f64=f64), torch.no_grad():
            mc = build_model("cpu", bias_shift=shift)
            b = batch
            if f64:
                mc.double()
                b = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
            wrap(mc, tag)
            fb = mc.forward_backbone

            def fb_w(batch_input, batch_size, want_preds=True):
                feats, sem, preds = fb(batch_input, batch_size, want_preds=True)
                own = sem.max(1)[1]
                diff = torch.nonzero(own != g_preds).view(-1)
                top2 = torch.sort(sem[diff].double(), dim=1)[0][:, -2:] if diff.numel() else torch.zeros(0, 2)
                flips[tag] = (int(diff.numel()), float((top2[:, 1] - top2[:, 0]).max()) if diff.numel() else 0.0)
                return feats, sem, g_preds  # the GPU's class decision (integers); features and scores stay this run's own

            mc.forward_backbone = fb_w
            np.random.seed(11)
            o = mc(b, 300, training=False)
        return mc, o

    mc32, o32 = host("f32", False)
    mc64, o64 = host("f64", True)
    for tag in ("f32", "f64"):  # differing class decisions are near-ties
        assert flips[tag][0] <= 8 and flips[tag][1] < 2e-4, flips
    # integers: identical in all three runs
    for o, mc, tag in ((o32, mc32, "f32"), (o64, mc64, "f64")):
        assert (o["fg_idxs"].numpy() == g_fg).all()
        assert (m.last_sampling_indices.cpu().numpy() == mc.last_sampling_indices.numpy()).all()
        assert (cap["gpu"]["pre_enc_inds"] == cap[tag]["pre_enc_inds"]).all()
        assert (cap["gpu"]["context_locs"] == cap[tag]["context_locs"]).all()
        assert (cap["gpu"]["geo"] == cap[tag]["geo"]).all()  # reach sets and fp32 sums (carried as doubles in the arbiter)
    log = []
    ok = True
    ref = {"semantic_scores": o64["semantic_scores"].numpy(), "context_feats": cap["f64"]["context_feats"],
           "dec": cap["f64"]["dec"][-1], "cls_logits": o64["mask_predictions"][-1]["cls_logits"].numpy(),
           "mask_logits": o64["mask_predictions"][-1]["mask_logits"][0].numpy()}
    runs = {"gpu": {"semantic_scores": g_sem, "context_feats": cap["gpu"]["context_feats"], "dec": cap["gpu"]["dec"][-1],
                    "cls_logits": out["mask_predictions"][-1]["cls_logits"].cpu().numpy(),
                    "mask_logits": out["mask_predictions"][-1]["mask_logits"][0].cpu().numpy()},
            "host_f32": {"semantic_scores": o32["semantic_scores"].numpy(), "context_feats": cap["f32"]["context_feats"],
                         "dec": cap["f32"]["dec"][-1], "cls_logits": o32["mask_predictions"][-1]["cls_logits"].numpy(),
                         "mask_logits": o32["mask_predictions"][-1]["mask_logits"][0].numpy()}}
    for run, d in runs.items():
        for k, v in d.items():
            ok = _arb_close(v, ref[k], f"{run}.{k}", log) and ok
    print("\narbiter (what, max-abs error vs float64, max|float64|, error in fp32 eps of that magnitude, bound):")
    for row in log:
        print("  %-28s %.3e %8.3f %6.1f %.3e" % row)
    assert ok, log
    # proposals: the same accepted set, scores to 1e-4
    pg, p64 = out["proposal_scores"], o64["proposal_scores"]
    assert len(pg[0]) == len(p64[0]) == len(o32["proposal_scores"][0])
    if len(pg[0]):
        assert (pg[0].cpu().numpy() == p64[0].numpy()).all()
        assert np.abs(pg[1].cpu().numpy() - p64[1].numpy()).max() < 1e-4


# ---- BASELINE config 4 at its real size -----------------------------------------------------------------------------
def test_fs_5shot_episode_full_size_matches_oracle_backend(hip, oracle):
    """1-way 5-shot as BASELINE.json names it, at scene size: an S150k query scene + five FULL support scenes of
    100k-140k points (process_support x 5 -> mean embedding -> GeoFormerFS.forward(..., training=False)).  The few-shot
    model samples ALL foreground points (no 50 000 cap) and 32 support centres: launch shapes the GeoFormer S150k test
    does not reach.  GPU episode against the same episode through the oracle's operators on the host; the host run is
    handed the GPU's class decision for the query scene after checking that its own differs on near-ties only."""
    from geoformer_amd import scene
    from geoformer_amd.model import GeoFormerFS, load_config
    from oracle import cpu_backend
    from oracle import oracle as orc
    from tests.util import synthetic_state_dict

    L = orc.lib()
    L.orc_set_threads.restype = int
    L.orc_set_threads(64)

    def dicts():
        q = scene.make_batch([scene.make_scene(150_000, 1234)])
        sups = [scene.make_batch([scene.make_scene(100_000 + 10_000 * i, 70 + i)]) for i in range(5)]
        for d in [q] + sups:
            d["batch_offsets"] = d["offsets"]
        for d in sups:
            d["support_masks"] = (d["instance_labels"] >= 0).long()
        return q, sups

    q, sups = dicts()

    def episode(device, preds=None):
        m = GeoFormerFS(load_config("test_geoformer_fs_scannet.yaml", k_shot=5))
        m.load_state_dict(synthetic_state_dict(m.state_dict(), 2))
        with torch.no_grad():
            m.semantic_linear.bias[3] += 1.0
        m.to(device)
        m.eval()
        mv = lambda d: {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in d.items()}  # noqa: E731
        cap, flips = [], {}
        orig = m.get_mask_prediction

        def gmp(*a, **k):
            r = orig(*a, **k)
            cap.append(r[-1]["mask_logits"][0].detach().cpu())
            return r

        m.get_mask_prediction = gmp
        np.random.seed(11)
        with torch.no_grad():
            embs = [m.process_support(mv(d), training=False) for d in sups]
            emb = torch.stack(embs).mean(0)
            fb = m.forward_backbone
            own = {}

            def fb_w(batch_input, batch_size):
                feats, sem, p = fb(batch_input, batch_size)
                own["preds"] = p.detach().cpu()
                if preds is not None:
                    diff = torch.nonzero(p.cpu() != preds).view(-1)
                    top2 = torch.sort(sem[diff].double(), dim=1)[0][:, -2:]
                    flips["n"], flips["gap"] = int(diff.numel()), float((top2[:, 1] - top2[:, 0]).max()) if diff.numel() else 0.0
                    p = preds.to(p.device)
                return feats, sem, p

            m.forward_backbone = fb_w
            out = m(None, mv(q), training=False, remember=False, support_embeddings=emb)
        scores, props = out["proposal_scores"]
        return dict(embs=torch.stack(embs).cpu(), sem=out["semantic_scores"].cpu(), fg=m.cache_data[3].cpu(),
                    inds=m.cache_data[2].cpu(), ml=cap[0], preds=own["preds"], flips=flips,
                    scores=scores.cpu() if len(scores) else torch.zeros(0),
                    npts=props.sum(1).cpu() if len(scores) else torch.zeros(0))

    got = episode("cuda")
    with cpu_backend.installed():
        ref = episode("cpu", preds=got["preds"])
    assert ref["flips"].get("n", 0) <= 8 and ref["flips"].get("gap", 0.0) < 2e-4, ref["flips"]
    assert (got["embs"] - ref["embs"]).abs().max() < 1e-4  # the five support embeddings
    assert _close_2fp32(got["sem"].numpy(), ref["sem"].numpy())  # (GPU against host: two fp32 paths)
    assert torch.equal(got["fg"], ref["fg"]) and got["fg"].numel() > 30_000
    assert torch.equal(got["inds"], ref["inds"])  # FPS over ALL foreground points of the query scene
    assert got["ml"].shape == ref["ml"].shape
    assert _close_2fp32(got["ml"].numpy(), ref["ml"].numpy())
    assert got["scores"].shape == ref["scores"].shape
    if len(ref["scores"]):
        assert (got["scores"] - ref["scores"]).abs().max() < 1e-4
        assert (got["npts"] - ref["npts"]).abs().max() <= 3
