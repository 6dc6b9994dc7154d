"""CPU: the sparse-conv oracle against dense torch convolutions on densified voxel sets.

spconv 1.0 is absent from the reference tree and un-tested there (parity unpinned); this
pins the oracle's restated semantics to the dense definition:
subm conv == zero-padded dense cross-correlation evaluated at active sites, strided conv ==
dense k2/s2 cross-correlation, inverse conv == transposed conv restricted to the fine sites.
"""
import numpy as np
import torch
import torch.nn.functional as F

from tests.util import random_voxels


def _densify(coords, feats, shape, batch):
    C = feats.shape[1]
    D = torch.zeros((batch, C) + tuple(shape), dtype=torch.float64)
    c = torch.from_numpy(coords).long()
    D[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]] = torch.from_numpy(feats).double()
    return D


def test_subm3_matches_dense(oracle):
    rng = np.random.default_rng(0)
    shape, B, Cin, Cout = (12, 10, 9), 2, 5, 7
    coords = random_voxels(rng, 300, shape, B, surface=False)
    M = coords.shape[0]
    feats = rng.standard_normal((M, Cin)).astype(np.float32)
    W = rng.standard_normal((27, Cin, Cout)).astype(np.float32)
    nbr = oracle.rules_subm3(coords, shape)
    assert (nbr[13, :M] == np.arange(M)).all()
    out = oracle.conv_fwd(feats, W, nbr, M)
    D = _densify(coords, feats, shape, B)
    w = torch.from_numpy(W).double().view(3, 3, 3, Cin, Cout).permute(4, 3, 0, 1, 2)
    ref = F.conv3d(D, w, padding=1)
    c = torch.from_numpy(coords).long()
    ref = ref[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]].numpy()
    assert np.abs(out - ref).max() < 1e-4


def test_down2_and_inverse_match_dense(oracle):
    rng = np.random.default_rng(1)
    shape, B, Cin, Cout = (11, 8, 13), 2, 4, 6  # odd dims: the last slab has no output cell
    coords = random_voxels(rng, 400, shape, B, surface=False)
    M = coords.shape[0]
    feats = rng.standard_normal((M, Cin)).astype(np.float32)
    W = rng.standard_normal((8, Cin, Cout)).astype(np.float32)
    out_coords, child, parent, koff = oracle.rules_down2(coords, shape)
    Mo = out_coords.shape[0]
    oshape = tuple((s - 2) // 2 + 1 for s in shape)
    # canonical order: ascending linear index
    lin = ((out_coords[:, 0].astype(np.int64) * oshape[0] + out_coords[:, 1]) * oshape[1] + out_coords[:, 2]) * oshape[2] + out_coords[:, 3]
    assert (np.diff(lin) > 0).all()
    # dropped inputs are exactly those on an odd max face
    dropped = parent < 0
    expect = (coords[:, 1] // 2 >= oshape[0]) | (coords[:, 2] // 2 >= oshape[1]) | (coords[:, 3] // 2 >= oshape[2])
    assert (dropped == expect).all() and dropped.any()
    out = oracle.conv_fwd(feats, W, child, Mo)
    D = _densify(coords, feats, shape, B)
    w = torch.from_numpy(W).double().view(2, 2, 2, Cin, Cout).permute(4, 3, 0, 1, 2)
    ref = F.conv3d(D, w, stride=2)
    assert tuple(ref.shape[2:]) == oshape
    oc = torch.from_numpy(out_coords).long()
    got_dense = torch.zeros_like(ref)
    got_dense[oc[:, 0], :, oc[:, 1], oc[:, 2], oc[:, 3]] = torch.from_numpy(out).double()
    assert (got_dense - ref).abs().max() < 1e-4  # also proves inactive outputs are exactly the empty cells

    # inverse conv: fine rows <- coarse rows through the one-hot table
    Wi = rng.standard_normal((8, Cout, Cin)).astype(np.float32)
    up = oracle.up_table(parent, koff)
    back = oracle.conv_fwd(out, Wi, up, M)
    wt = torch.from_numpy(Wi).double().view(2, 2, 2, Cout, Cin).permute(3, 4, 0, 1, 2)
    refT = F.conv_transpose3d(got_dense, wt, stride=2)
    c = torch.from_numpy(coords).long()
    keep = ~torch.from_numpy(dropped)
    r = refT[c[keep, 0], :, c[keep, 1], c[keep, 2], c[keep, 3]].numpy()
    assert np.abs(back[keep.numpy()] - r).max() < 1e-4
    assert (back[dropped] == 0).all()


def test_conv_grads_match_autograd(oracle):
    rng = np.random.default_rng(2)
    shape, B, Cin, Cout = (9, 9, 9), 1, 3, 4
    coords = random_voxels(rng, 150, shape, B, surface=False)
    M = coords.shape[0]
    feats = rng.standard_normal((M, Cin)).astype(np.float32)
    W = rng.standard_normal((27, Cin, Cout)).astype(np.float32)
    dout = rng.standard_normal((M, Cout)).astype(np.float32)
    nbr = oracle.rules_subm3(coords, shape)
    ft = torch.from_numpy(feats).double().requires_grad_()
    Wt = torch.from_numpy(W).double().requires_grad_()
    out = torch.zeros(M, Cout, dtype=torch.float64)
    for k in range(27):
        o = np.nonzero(nbr[k, :M] >= 0)[0]
        i = nbr[k, o]
        out = out.index_add(0, torch.from_numpy(o), ft[torch.from_numpy(i).long()] @ Wt[k])
    out.backward(torch.from_numpy(dout).double())
    assert np.abs(oracle.conv_dgrad(dout, W, nbr, M) - ft.grad.numpy()).max() < 1e-4
    assert np.abs(oracle.conv_wgrad(feats, dout, nbr, 27) - Wt.grad.numpy()).max() < 1e-4
