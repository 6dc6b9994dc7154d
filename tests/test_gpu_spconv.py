"""GPU parity: HIP rulebooks (bit-exact) and gather-MFMA convolution (<=1e-4 abs) vs the oracle."""
import numpy as np
import pytest
import torch

from tests.util import random_voxels

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("M,shape,B,surface", [(5000, (64, 48, 40), 2, True), (300, (12, 10, 9), 1, False),
                                               (17, (128, 128, 128), 1, False), (1, (128, 128, 128), 1, False)])
def test_subm_rules_bit_exact(hip, oracle, M, shape, B, surface):
    from geoformer_amd import sparse

    rng = np.random.default_rng(M)
    coords = random_voxels(rng, M, shape, B, surface)
    M = coords.shape[0]
    ref = oracle.rules_subm3(coords, shape)
    c = _dev(coords)
    ix = sparse.build_index(c, B, shape)
    rules = sparse.subm_rules(c, ix)
    got = rules.nbr.cpu().numpy()
    assert got.shape == ref.shape
    assert (got == ref).all()
    # group masks = OR of the offsets present in each 16-row group
    gm = rules.gmask.cpu().numpy().view(np.uint32)
    present = (ref >= 0).reshape(27, -1, 16).any(2)
    expect = (present * (1 << np.arange(27))[:, None]).sum(0).astype(np.uint32)
    assert (gm == expect).all()
    # canonical pair lists: ascending output row inside every offset
    for k, p in enumerate(rules.pairs()):
        p = p.cpu().numpy()
        assert (np.diff(p[1]) > 0).all()
        assert (ref[k, p[1]] == p[0]).all()


@pytest.mark.parametrize("M,shape,B", [(6000, (65, 47, 41), 2), (200, (11, 8, 13), 2), (3, (128, 128, 128), 1)])
def test_down_rules_bit_exact(hip, oracle, M, shape, B):
    from geoformer_amd import sparse

    rng = np.random.default_rng(M + 1)
    coords = random_voxels(rng, M, shape, B, surface=M > 1000)
    M = coords.shape[0]
    oc, child, parent, koff = oracle.rules_down2(coords, shape)
    r = sparse.down_rules(_dev(coords), B, shape)
    assert r.M_out == oc.shape[0]
    assert (r.out_coords.cpu().numpy() == oc).all()
    assert (r.child.cpu().numpy()[:, : child.shape[1]] == child).all()
    assert (r.parent.cpu().numpy() == parent).all()
    assert (r.koff.cpu().numpy() == koff).all()
    assert (r.up.cpu().numpy() == oracle.up_table(parent, koff, r.ld_up)).all()
    # the output index answers subm lookups of the next level without a permutation
    nxt = sparse.subm_rules(r.out_coords.contiguous(), r.index_out)
    assert (nxt.nbr.cpu().numpy() == oracle.rules_subm3(oc, r.out_shape)).all()


@pytest.mark.parametrize("M,shape,B,nl", [(20000, (150, 90, 64), 1, 6), (3000, (65, 47, 41), 2, 4), (5, (128, 128, 128), 1, 6)])
def test_down_rules_chain_bit_exact(hip, oracle, M, shape, B, nl):
    """gf_rules_down2_chain (all levels in one call, one workspace, device-side voxel counts) level by level against
    the oracle: coarse coordinates, child / parent / up tables, group masks, and the index each level hands to the
    submanifold rulebook of the next."""
    from geoformer_amd import sparse

    rng = np.random.default_rng(M + 7)
    coords = random_voxels(rng, M, shape, B, surface=M > 1000)
    chain = sparse.down_rules_chain(_dev(coords), B, shape, nl)
    cur, cur_shape = coords, tuple(shape)
    assert len(chain) >= 1
    for r in chain:
        oc, child, parent, koff = oracle.rules_down2(cur, cur_shape)
        assert r.M_in == cur.shape[0] and r.M_out == oc.shape[0]
        assert (r.out_coords.cpu().numpy() == oc).all()
        got_child = r.child.cpu().numpy()
        mo = oc.shape[0]  # the tables keep their capacity as leading dimension: rows beyond M_out stay empty
        assert (got_child[:, :mo] == child[:, :mo]).all() and (got_child[:, mo:] == -1).all() and (child[:, mo:] == -1).all()
        assert (r.parent.cpu().numpy() == parent).all() and (r.koff.cpu().numpy() == koff).all()
        assert (r.up.cpu().numpy() == oracle.up_table(parent, koff, r.ld_up)).all()
        for tbl, gm in ((got_child, r.gmask_down), (r.up.cpu().numpy(), r.gmask_up)):
            present = (tbl >= 0).reshape(8, -1, 16).any(2)
            expect = (present * (1 << np.arange(8))[:, None]).sum(0).astype(np.uint32)
            assert (gm.cpu().numpy().view(np.uint32) == expect).all()
        nxt = sparse.subm_rules(r.out_coords.contiguous(), r.index_out)
        assert (nxt.nbr.cpu().numpy() == oracle.rules_subm3(oc, r.out_shape)).all()
        cur, cur_shape = oc, tuple(r.out_shape)


def test_index_build_any_row_order(hip, oracle):
    """The bitmap bits are OR-ed across runs of lanes before the atomic: rows in raster order (long runs inside a
    word), shuffled (no runs) and with every word boundary crossed must give the same index."""
    from geoformer_amd import sparse

    rng = np.random.default_rng(5)
    shape = (40, 33, 70)
    dense = np.stack(np.meshgrid(np.arange(8, 20), np.arange(5, 25), np.arange(0, 70), indexing="ij"), -1).reshape(-1, 3)
    coords = np.concatenate([np.zeros((dense.shape[0], 1), np.int64), dense], 1).astype(np.int32)
    ref = oracle.rules_subm3(coords, shape)
    for order in (np.arange(coords.shape[0]), rng.permutation(coords.shape[0])):
        c = _dev(coords[order])
        rules = sparse.subm_rules(c, sparse.build_index(c, 1, shape))
        got = rules.nbr.cpu().numpy()[:, : coords.shape[0]]
        inv = np.empty_like(order); inv[order] = np.arange(order.size)
        expect = np.where(ref[:, order] >= 0, inv[np.clip(ref[:, order], 0, None)], -1)
        assert (got == expect).all()


@pytest.mark.parametrize("Cin,Cout", [(6, 16), (16, 16), (32, 16), (32, 32), (48, 64), (112, 112), (19, 21), (80, 160)])
def test_subm_conv_parity(hip, oracle, Cin, Cout):
    from geoformer_amd import sparse

    rng = np.random.default_rng(Cin * 131 + Cout)
    shape, B = (40, 36, 30), 2
    coords = random_voxels(rng, 3000, shape, B, surface=True)
    M = coords.shape[0]
    feats = rng.standard_normal((M, Cin)).astype(np.float32)
    W = (rng.standard_normal((27, Cin, Cout)) / np.sqrt(27 * Cin)).astype(np.float32)
    nbr = oracle.rules_subm3(coords, shape)
    ref = oracle.conv_fwd(feats, W, nbr, M)
    c = _dev(coords)
    rules = sparse.subm_rules(c, sparse.build_index(c, B, shape))
    out = sparse.conv_fwd(_dev(feats), _dev(W), rules.nbr, rules.gmask, 27, M, rules.ld)
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-4  # tolerance of BASELINE.json north_star


def test_conv_fused_bn_relu_residual_and_identity(hip, oracle):
    from geoformer_amd import sparse

    rng = np.random.default_rng(5)
    shape, B, Cin, Cout = (30, 30, 30), 1, 32, 16
    coords = random_voxels(rng, 2000, shape, B, surface=True)
    M = coords.shape[0]
    feats = rng.standard_normal((M, Cin)).astype(np.float32)
    W = (rng.standard_normal((27, Cin, Cout)) / np.sqrt(27 * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cin).astype(np.float32)
    shift = rng.standard_normal(Cin).astype(np.float32)
    res = rng.standard_normal((M, Cout)).astype(np.float32)
    nbr = oracle.rules_subm3(coords, shape)
    act = np.maximum(feats * scale + shift, 0).astype(np.float32)
    ref = oracle.conv_fwd(act, W, nbr, M) + res
    c = _dev(coords)
    rules = sparse.subm_rules(c, sparse.build_index(c, B, shape))
    out = sparse.conv_fwd(_dev(feats), _dev(W), rules.nbr, rules.gmask, 27, M, rules.ld, in_scale=_dev(scale),
                          in_shift=_dev(shift), residual=_dev(res))
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-4
    # K == 1 identity map (the 1x1x1 i_branch conv, geoformer_modules.py:17-19) == plain GEMM
    W1 = (rng.standard_normal((1, Cin, Cout)) / np.sqrt(Cin)).astype(np.float32)
    out1 = sparse.conv_fwd(_dev(feats), _dev(W1), None, None, 1, M, 0)
    assert np.abs(out1.cpu().numpy() - feats @ W1[0]).max() < 1e-4


def test_down_up_conv_parity(hip, oracle):
    from geoformer_amd import sparse

    rng = np.random.default_rng(9)
    shape, B, C0, C1 = (41, 36, 31), 2, 16, 32
    coords = random_voxels(rng, 4000, shape, B, surface=True)
    M = coords.shape[0]
    feats = rng.standard_normal((M, C0)).astype(np.float32)
    Wd = (rng.standard_normal((8, C0, C1)) / np.sqrt(8 * C0)).astype(np.float32)
    Wu = (rng.standard_normal((8, C1, C0)) / np.sqrt(C1)).astype(np.float32)
    oc, child, parent, koff = oracle.rules_down2(coords, shape)
    ref_d = oracle.conv_fwd(feats, Wd, child, oc.shape[0])
    ref_u = oracle.conv_fwd(ref_d, Wu, oracle.up_table(parent, koff), M)
    r = sparse.down_rules(_dev(coords), B, shape)
    d = sparse.conv_fwd(_dev(feats), _dev(Wd), r.child, r.gmask_down, 8, r.M_out, r.ld)
    u = sparse.conv_fwd(d, _dev(Wu), r.up, r.gmask_up, 8, M, r.ld_up)
    assert np.abs(d.cpu().numpy() - ref_d).max() < 1e-4
    assert np.abs(u.cpu().numpy() - ref_u).max() < 1e-4


def test_spconv_modules_forward_backward(hip, oracle):
    """The drop-in spconv modules with autograd: subm -> down -> subm -> inverse, gradients of the input
    features and of every weight against the oracle's dgrad/wgrad restatement."""
    from geoformer_amd import spconv

    rng = np.random.default_rng(21)
    shape, B, C0, C1 = (37, 33, 30), 2, 16, 32
    coords = random_voxels(rng, 3000, shape, B, surface=True)
    M = coords.shape[0]
    feats = rng.standard_normal((M, C0)).astype(np.float32)
    mods = [spconv.SubMConv3d(C0, C0, 3, padding=1, bias=False, indice_key="subm1"),
            spconv.SparseConv3d(C0, C1, kernel_size=2, stride=2, bias=False, indice_key="spconv1"),
            spconv.SubMConv3d(C1, C1, 3, padding=1, bias=False, indice_key="subm2"),
            spconv.SparseInverseConv3d(C1, C0, kernel_size=2, bias=False, indice_key="spconv1")]
    for m in mods:
        m.cuda()
    x = torch.from_numpy(feats).cuda().requires_grad_()
    t = spconv.SparseConvTensor(x, _dev(coords), shape, B)
    for m in mods:
        t = m(t)
    gout = rng.standard_normal((M, C0)).astype(np.float32)
    t.features.backward(_dev(gout))

    W = [m.weight.detach().cpu().numpy().reshape(-1, m.in_channels, m.out_channels) for m in mods]
    nbr1 = oracle.rules_subm3(coords, shape)
    oc, child, parent, koff = oracle.rules_down2(coords, shape)
    up = oracle.up_table(parent, koff)
    nbr2 = oracle.rules_subm3(oc, tuple((s - 2) // 2 + 1 for s in shape))
    y1 = oracle.conv_fwd(feats, W[0], nbr1, M)
    y2 = oracle.conv_fwd(y1, W[1], child, oc.shape[0])
    y3 = oracle.conv_fwd(y2, W[2], nbr2, oc.shape[0])
    y4 = oracle.conv_fwd(y3, W[3], up, M)
    assert np.abs(t.features.detach().cpu().numpy() - y4).max() < 1e-4
    g3 = oracle.conv_dgrad(gout, W[3], up, oc.shape[0])
    g2 = oracle.conv_dgrad(g3, W[2], nbr2, oc.shape[0])
    g1 = oracle.conv_dgrad(g2, W[1], child, M)
    g0 = oracle.conv_dgrad(g1, W[0], nbr1, M)
    scale = max(1.0, float(np.abs(g0).max()))
    assert np.abs(x.grad.cpu().numpy() - g0).max() < 1e-4 * scale
    dWs = [oracle.conv_wgrad(feats, g1, nbr1, 27), oracle.conv_wgrad(y1, g2, child, 8),
           oracle.conv_wgrad(y2, g3, nbr2, 27), oracle.conv_wgrad(y3, gout, up, 8)]
    for m, dW in zip(mods, dWs):
        got = m.weight.grad.cpu().numpy().reshape(dW.shape)
        assert np.abs(got - dW).max() < 2e-4 * max(1.0, float(np.abs(dW).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("M,Cin,Cout", [(50_000, 32, 16), (3_001, 224, 112), (17, 64, 32)])
def test_wgrad_without_table_is_xt_g(hip, M, Cin, Cout):
    """gf_conv_wgrad_masked(_acc) with K = 1 and no table (a 1x1x1 convolution: the training executor's identity
    branches): dW = X^T G on the tiled kernel, against float64; the _acc form adds to what dW holds."""
    from geoformer_amd import _lib
    from geoformer_amd._lib import check, ptr, stream_ptr

    g = torch.Generator(device="cuda").manual_seed(M)
    x = torch.randn(M, Cin, device="cuda", generator=g)
    gy = torch.randn(M, Cout, device="cuda", generator=g)
    want = (x.double().t() @ gy.double())
    lib = _lib.load()
    dW = torch.full((1, Cin, Cout), 7.0, device="cuda")
    check(lib.gf_conv_wgrad_masked(ptr(x), ptr(gy), None, None, 1, M, 0, Cin, Cout, ptr(dW), stream_ptr()), "wgrad")
    tol = 1e-5 * float(want.abs().max()) + 1e-4
    assert float((dW[0].double() - want).abs().max()) <= tol
    check(lib.gf_conv_wgrad_masked_acc(ptr(x), ptr(gy), None, None, 1, M, 0, Cin, Cout, ptr(dW), stream_ptr()), "wgrad")
    assert float((dW[0].double() - 2 * want).abs().max()) <= 2 * tol


def _flat_reference(nbr, M, K=27):
    """The flat step table's step records restated in numpy: per 16-row group the present offsets, ascending."""
    ng = (M + 15) // 16
    pad = np.full((K, ng * 16), -1, np.int32)
    pad[:, :M] = nbr[:, :M]
    recs, goff, masks = [], [0], []
    for g in range(ng):
        blk = pad[:, g * 16:(g + 1) * 16]
        ks = [k for k in range(K) if (blk[k] >= 0).any()]
        masks.append(sum(1 << k for k in ks))
        recs += [blk[k] for k in ks]
        goff.append(goff[-1] + len(ks))
    return np.array(recs, np.int32).reshape(-1, 16), np.array(goff), np.array(masks, np.uint32)


@pytest.mark.parametrize("M,shape,nbins", [(3000, (40, 36, 30), 0), (30000, (90, 80, 60), 0), (777, (20, 22, 18), 64), (40, (8, 8, 8), 0)])
def test_flat_step_table(hip, oracle, M, shape, nbins):
    """gf_rules_flat_steps against the neighbour table it restates: step records and per-group offsets bit for bit;
    the bin table: every group exactly once, sizes descending along the sorted positions, snake order, bins balanced."""
    from geoformer_amd import sparse

    rng = np.random.default_rng(M)
    coords = random_voxels(rng, M, shape, 1, surface=True)
    M = coords.shape[0]
    nbr = oracle.rules_subm3(coords, shape)
    c = _dev(coords)
    rules = sparse.subm_rules(c, sparse.build_index(c, 1, shape))
    flat = sparse.flat_steps(rules.nbr, rules.gmask, 27, M, rules.ld, nbins).cpu().numpy()
    recs, goff, masks = _flat_reference(nbr, M)
    ng = (M + 15) // 16
    S = int(goff[-1])
    nb = nbins if nbins else 1024
    rounds = (ng + nb - 1) // nb
    assert tuple(flat[:5]) == (S, nb, ng, 27, rounds)
    assert (rules.gmask.cpu().numpy().view(np.uint32)[:ng] == masks).all()
    sizes = np.diff(goff)
    assert (flat[8: 8 + 32] == np.bincount(sizes, minlength=32)).all()
    assert (flat[128: 128 + ng + 1] == goff).all()
    ppos = flat[128 + ng + 1: 128 + 2 * ng + 1]
    assert (np.sort(ppos) == np.arange(ng)).all()
    d_at = (128 + 2 * ng + 1 + 63) // 64 * 64
    desc = flat[d_at: d_at + (rounds + 1) * nb * 4].reshape(rounds + 1, nb, 4)
    s_at = d_at + (rounds + 1) * nb * 4
    got = flat[s_at: s_at + (S + 32) * 16].reshape(-1, 16)
    assert (got[:S] == recs).all()
    assert (got[S:] == -1).all()
    # sorted positions: round j runs over the bins forwards (even) or backwards (odd)
    order = np.concatenate([desc[j] if j % 2 == 0 else desc[j, ::-1] for j in range(rounds + 1)])
    filled = order[:, 0] >= 0
    assert filled[:ng].all() and not filled[ng:].any()
    gs = order[:ng, 0]
    assert (np.sort(gs) == np.arange(ng)).all()
    assert (order[:ng, 1] == goff[gs]).all() and (order[:ng, 2] == sizes[gs]).all()
    assert (order[:ng, 3].view(np.uint32) == masks[gs]).all()
    assert (np.diff(order[:ng, 2]) <= 0).all()
    load = np.where(desc[:, :, 0] >= 0, desc[:, :, 2], 0).sum(0)
    assert load.max() - load.min() <= 27


@pytest.mark.parametrize("Cin,Cout", [(16, 16), (32, 16), (16, 32), (32, 32), (64, 32), (48, 32), (64, 16)])
def test_lds_weight_conv_parity(hip, oracle, Cin, Cout):
    """k_conv_lw (forced through the dev knob on a small input) over the flat step table: every combination of fused
    prologue / residual / epilogue activation / second output against the oracle."""
    from geoformer_amd import _lib, sparse
    from geoformer_amd._lib import ptr, stream_ptr

    rng = np.random.default_rng(Cin * 31 + Cout)
    shape, B = (40, 36, 30), 2
    coords = random_voxels(rng, 3000, shape, B, surface=True)
    M = coords.shape[0]
    feats = rng.standard_normal((M, Cin)).astype(np.float32)
    W = (rng.standard_normal((27, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cin).astype(np.float32)
    shift = (0.3 * rng.standard_normal(Cin)).astype(np.float32)
    osc = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    osh = (0.3 * rng.standard_normal(Cout)).astype(np.float32)
    res = rng.standard_normal((M, Cout)).astype(np.float32)
    nbr = oracle.rules_subm3(coords, shape)
    c = _dev(coords)
    rules = sparse.subm_rules(c, sparse.build_index(c, B, shape))
    flat = sparse.flat_steps(rules.nbr, rules.gmask, 27, M, rules.ld)
    x, w, s, t, r, a, b = (_dev(v) for v in (feats, W, scale, shift, res, osc, osh))
    act_in = np.maximum(feats * scale + shift, 0).astype(np.float32)
    try:
        sparse.dev_conv_knobs(lw=1)
        for aff in (False, True):
            for resid in (False, True):
                ref = oracle.conv_fwd(act_in if aff else feats, W, nbr, M) + (res if resid else 0)
                kw = dict(in_scale=s, in_shift=t) if aff else {}
                if resid:
                    kw["residual"] = r
                got = sparse.conv_fwd(x, w, rules.nbr, rules.gmask, 27, M, rules.ld, flat=flat, **kw)
                assert np.abs(got.cpu().numpy() - ref).max() < 1e-4, (aff, resid)
                got = sparse.conv_fwd(x, w, rules.nbr, rules.gmask, 27, M, rules.ld, flat=flat, out_scale=a, out_shift=b, **kw)
                assert np.abs(got.cpu().numpy() - np.maximum(ref * osc + osh, 0)).max() < 1e-4, (aff, resid, "act")
        # both outputs (the raw sums and the activated copy) in one launch
        lib = _lib.load()
        wp = sparse.pack_weights(w)
        out = torch.empty(M, Cout, device="cuda")
        out_act = torch.empty(M, Cout, device="cuda")
        sparse.check(lib.gf_conv_fwd_flat(ptr(x), ptr(wp), ptr(rules.nbr), ptr(rules.gmask), None, ptr(flat), 27, M, M, rules.ld,
                                          Cin, Cout, None, None, ptr(r), ptr(a), ptr(b), ptr(out), ptr(out_act), stream_ptr()),
                     "gf_conv_fwd_flat")
        ref = oracle.conv_fwd(feats, W, nbr, M) + res
        assert np.abs(out.cpu().numpy() - ref).max() < 1e-4
        assert np.abs(out_act.cpu().numpy() - np.maximum(ref * osc + osh, 0)).max() < 1e-4
    finally:
        sparse.dev_conv_knobs()
