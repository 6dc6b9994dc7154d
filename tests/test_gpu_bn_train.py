"""Fused training-mode BatchNorm1d + ReLU over voxel rows (csrc/bn_train.hip) against nn.BatchNorm1d + ReLU evaluated in
float64 on the host: output, running statistics, input / weight / bias gradients; the widths and row counts of the
U-Net's levels, a two-row batch, a channel whose mean dwarfs its spread (no E[x^2] - mean^2 cancellation), and the
module route: spconv.SparseSequential takes the fused pair in training mode and the framework's kernels in eval mode."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, w, b, rm, rv, gy, eps=1e-4, momentum=0.1):
    bn = torch.nn.BatchNorm1d(x.shape[1], eps=eps, momentum=momentum).double()
    with torch.no_grad():
        bn.weight.copy_(w); bn.bias.copy_(b); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    xd = x.double().clone().requires_grad_()
    z = bn(xd)
    y = torch.relu(z)
    y.backward(gy.double())
    _ref.pre = z.detach()
    return y.detach(), xd.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()


@pytest.mark.parametrize("M,C", [(100_003, 16), (5000, 224), (37, 112), (2, 16), (754, 80), (523_000, 32)])
def test_bn_relu_train_matches_float64(hip, M, C):
    from geoformer_amd import pointops

    rng = np.random.default_rng(M + C)
    x = torch.from_numpy((rng.standard_normal((M, C)) * rng.uniform(0.5, 3.0, C) + rng.normal(0, 2.0, C)).astype(np.float32))
    x[:, 0] += 100.0  # |mean| >> std: raw second moments would lose the variance's digits in fp32
    w = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32))
    b = torch.from_numpy(rng.normal(0, 0.3, C).astype(np.float32))
    rm = torch.from_numpy(rng.normal(0, 0.1, C).astype(np.float32))
    rv = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32))
    gy = torch.from_numpy(rng.standard_normal((M, C)).astype(np.float32))
    y_r, gx_r, gw_r, gb_r, rm_r, rv_r = _ref(x, w, b, rm, rv, gy)
    bn = torch.nn.BatchNorm1d(C, eps=1e-4, momentum=0.1).cuda()
    with torch.no_grad():
        bn.weight.copy_(w); bn.bias.copy_(b); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.train()
    xg = x.cuda().requires_grad_()
    assert pointops.bn_relu_train_supported(bn, xg)
    y = pointops.bn_relu_train(bn, xg)
    y.backward(gy.cuda())
    torch.cuda.synchronize()
    assert int(bn.num_batches_tracked) == 1
    c = lambda t: t.detach().cpu().double()  # noqa: E731
    assert (c(y) - y_r).abs().max() < 2e-5 * max(1.0, float(y_r.abs().max()))
    assert (c(bn.running_mean) - rm_r).abs().max() < 1e-5 * max(1.0, float(rm_r.abs().max()))
    assert (c(bn.running_var) - rv_r).abs().max() < 1e-4 * max(1.0, float(rv_r.abs().max()))
    s = max(1.0, float(gx_r.abs().max()))
    # (elements whose pre-activation is within fp32 rounding of zero may take the other side of the ReLU: the channel
    # with |mean| >> std resolves xhat to ~2e-5 only; those single elements are not compared)
    clear = _ref.pre.abs() > 1e-3
    assert ((c(xg.grad) - gx_r).abs() * clear).max() < 1e-4 * s
    assert float((~clear).double().mean()) < 2e-3
    for got, ref in ((bn.weight.grad, gw_r), (bn.bias.grad, gb_r)):
        # (sums over M rows of O(1) terms; a handful of ReLU ties on the other side move them by O(1e-2))
        assert (c(got) - ref).abs().max() < 1e-4 * max(1.0, float(ref.abs().max()), np.sqrt(M)), (got, ref)
    # twice in a row on the same stream: the arrival counter is back at zero
    y2 = pointops.bn_relu_train(bn, x.cuda())
    torch.cuda.synchronize()
    assert torch.equal(y2, y.detach())


def test_sparse_sequential_takes_the_fused_pair_in_training(hip):
    from geoformer_amd import spconv
    from geoformer_amd.model.layers import BatchNorm1d

    torch.manual_seed(0)
    seq = spconv.SparseSequential(BatchNorm1d(32, eps=1e-4, momentum=0.1), torch.nn.ReLU()).cuda()
    ref = torch.nn.Sequential(torch.nn.BatchNorm1d(32, eps=1e-4, momentum=0.1), torch.nn.ReLU()).cuda()
    ref.load_state_dict({k.replace("0.", "0."): v for k, v in seq.state_dict().items()})
    x = torch.randn(3000, 32, device="cuda") * 2 + 1
    coords = torch.zeros(3000, 4, dtype=torch.int32, device="cuda")
    for mode in (True, False):
        seq.train(mode); ref.train(mode)
        t = spconv.SparseConvTensor(x.clone().requires_grad_(), coords, [8, 8, 8], 1)
        xin = t.features
        out = seq(t).features
        xr = x.clone().requires_grad_()
        outr = ref(xr)
        out.sum().backward(); outr.sum().backward()
        assert (out - outr).abs().max() < 1e-5 and (xin.grad - xr.grad).abs().max() < 1e-5
    assert (seq[0].running_mean - ref[0].running_mean).abs().max() < 1e-6
    assert (seq[0].running_var - ref[0].running_var).abs().max() < 1e-5
    assert int(seq.state_dict()["0.num_batches_tracked"]) == int(ref.state_dict()["0.num_batches_tracked"]) == 1


@pytest.mark.parametrize("shape", [(1, 16, 100_003), (4, 32, 2048, 64), (2, 64, 16), (1, 16, 37)])
def test_bn_train_channel_major_matches_float64(hip, shape):
    """The channel-major pair (BatchNorm1d over [B,C,L], BatchNorm2d over [B,C,H,W]) through the model's own layer
    classes against nn.BatchNorm in float64: output, running statistics, gradients; L not a multiple of 4."""
    from geoformer_amd.model.layers import BatchNorm1d, BatchNorm2d

    C = shape[1]
    rng = np.random.default_rng(sum(shape))
    x = torch.from_numpy((rng.standard_normal(shape) * 2.0 + 0.7).astype(np.float32))
    x[:, 0] += 100.0
    gy = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    cls, ref_cls = (BatchNorm2d, torch.nn.BatchNorm2d) if len(shape) == 4 else (BatchNorm1d, torch.nn.BatchNorm1d)
    bn, ref = cls(C, eps=1e-5, momentum=0.1).cuda(), ref_cls(C, eps=1e-5, momentum=0.1).double()
    w = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32))
    b = torch.from_numpy(rng.normal(0, 0.3, C).astype(np.float32))
    with torch.no_grad():
        bn.weight.copy_(w); bn.bias.copy_(b); ref.weight.copy_(w); ref.bias.copy_(b)
    bn.train(); ref.train()
    xg = x.cuda().requires_grad_()
    y = bn(xg)
    y.backward(gy.cuda())
    xr = x.double().requires_grad_()
    yr = ref(xr)
    yr.backward(gy.double())
    torch.cuda.synchronize()
    c = lambda t: t.detach().cpu().double()  # noqa: E731
    assert (c(y) - yr.detach()).abs().max() < 2e-5 * max(1.0, float(yr.abs().max()))
    assert (c(xg.grad) - xr.grad).abs().max() < 1e-4 * max(1.0, float(xr.grad.abs().max()))
    n = x.numel() // C
    for got, r in ((bn.weight.grad, ref.weight.grad), (bn.bias.grad, ref.bias.grad)):
        assert (c(got) - r).abs().max() < 2e-5 * max(1.0, float(r.abs().max()), np.sqrt(n))
    sd = bn.state_dict()
    assert (c(sd["running_mean"]) - ref.running_mean).abs().max() < 1e-5 * max(1.0, float(ref.running_mean.abs().max()))
    assert (c(sd["running_var"]) - ref.running_var).abs().max() < 1e-4 * max(1.0, float(ref.running_var.abs().max()))
    assert int(sd["num_batches_tracked"]) == 1
