"""The multi-GPU path EXECUTED on the one GPU a test box has: an `nccl` (= RCCL) process group of world size 1.

  reference: /root/reference/train.py:156-185 (the intended DDP + SyncBatchNorm launch)

The group lives in the pytest process itself (no child process: nothing is exec'ed from a process that holds the GPU),
is created by a module fixture and destroyed after the module.  With one rank every collective is the identity, so
the forced-exchange step must reproduce the plain step: what these tests add is that librccl loads, the communicator
comes up on an MI355X, the reducer's hooks / pack / re-point / asynchronous all-reduce sequence and SyncBatchNorm1d's
packed all-gather / all-reduce run on DEVICE tensors through it.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.fixture(scope="module")
def rccl_group(hip):
    from geoformer_amd import parallel

    assert not dist.is_initialized()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    saved = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE")}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    assert parallel.init_distributed("nccl", force=True) == 1
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    yield parallel
    parallel.FORCE_EXCHANGE = False
    torch.cuda.synchronize()
    dist.destroy_process_group()
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_rccl_collectives_on_device(rccl_group):
    x = torch.arange(1024, dtype=torch.float32, device="cuda") * 0.5
    y = x.clone()
    w = dist.all_reduce(y, async_op=True)
    w.wait()
    assert torch.equal(x, y)
    out = torch.empty(1024, dtype=torch.float32, device="cuda")
    dist.all_gather_into_tensor(out, x)
    assert torch.equal(out, x)
    t = torch.tensor([3.0], device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.broadcast(t, 0)
    dist.barrier()
    torch.cuda.synchronize()
    assert float(t.item()) == 3.0
    assert rccl_group.max_over_ranks(1.25, torch.device("cuda", 0)) == 1.25
    assert rccl_group.all_ranks_agree(True, torch.device("cuda", 0)) is True
    with open("/proc/self/maps") as f:
        maps = f.read()
    assert "librccl" in maps, "the nccl backend of this PyTorch must be RCCL"


def test_reducer_forced_exchange_toy(rccl_group):
    """Hooks / pack / re-point / all-reduce on device tensors: same gradients as plain autograd, unused parameters get
    grad None back, buckets start inside the backward."""
    parallel = rccl_group
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3)).cuda()
    unused = torch.nn.Linear(4, 4).cuda()
    mod = torch.nn.ModuleList([unused, net])
    red = parallel.BucketedGradReducer(mod, bucket_bytes=64, always_exchange=True)
    assert red.exchanging() and len(red.ranges) >= 3
    ref = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3)).cuda()
    ref.load_state_dict(net.state_dict())
    for it in range(2):
        x = torch.randn(6, 5, device="cuda")
        red.prepare()
        assert red._hooks, "forced exchange keeps the gradient hooks"
        net(x).square().sum().backward()
        assert red.launched_in_backward >= 1
        red.finish()
        ref.zero_grad(set_to_none=True)
        ref(x).square().sum().backward()
        for p, q in zip(net.parameters(), ref.parameters()):
            assert p.grad is not None and p.grad.data_ptr() != 0
            # the gradient now lives in the flat exchange buffer
            lo, hi = red.flat.data_ptr(), red.flat.data_ptr() + red.flat.numel() * 4
            assert lo <= p.grad.data_ptr() < hi
            assert torch.equal(p.grad, q.grad)
        assert all(p.grad is None for p in unused.parameters())
    # a plain reducer in the same one-rank group does nothing of the kind
    plain = parallel.BucketedGradReducer(mod, bucket_bytes=64)
    assert not plain.exchanging() and not plain._hooks


def test_train_dp_step_forced_exchange_matches_plain_step(rccl_group):
    """tools/train_dp.step on the real model (small scene, dropout 0): reducer forced on == plain step -- loss, every
    gradient and the Adam update (atomics in the backward: compared within fp32 round-off, not bit for bit)."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import train_dp

    parallel = rccl_group
    dev = torch.device("cuda", 0)
    res = {}
    for forced in (False, True):
        args = train_dp.default_args(small=True, points=8192, batch_size=2, epoch=200, prepare_epochs=100,
                                     always_exchange=forced)
        cfg, m, crit = train_dp.build(args, dev)
        red = parallel.BucketedGradReducer(m, bucket_bytes=1 << 20, always_exchange=forced)
        assert red.exchanging() == forced
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, fused=True)
        batch = train_dp.make_batches(args, 0, dev, 1)[0]
        loss, info = train_dp.step(m, crit, red, None, batch, args.epoch, 7)
        grads = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m.named_parameters()}
        opt.step()
        torch.cuda.synchronize()
        res[forced] = (loss, grads, {n: p.detach().clone() for n, p in m.named_parameters()}, red.launched_in_backward)
    (l0, g0, p0, _), (l1, g1, p1, early) = res[False], res[True]
    assert np.isfinite(l0) and abs(l0 - l1) <= 1e-5 * max(1.0, abs(l0))
    # (buckets leave in index order: whether any starts inside the backward depends on bucket 0's parameters all
    #  receiving a gradient -- covered by the toy test above; here every bucket must have gone through the exchange)
    assert early >= 0
    n_with = 0
    gmax = max(float(g.abs().max()) for g in g0.values() if g is not None)
    for n in g0:
        assert (g0[n] is None) == (g1[n] is None), n
        if g0[n] is None:
            continue
        n_with += 1
        scale = float(g0[n].abs().max()) + 1e-12
        # (a gradient that is zero in exact arithmetic -- the key bias of a softmax attention -- is round-off on both
        #  sides: bounded against the step's largest gradient instead of against itself)
        assert float((g0[n] - g1[n]).abs().max()) <= 2e-4 * scale + 1e-6 * gmax, n
        # the first Adam step moves a weight by lr * g / (|g| + eps): equal wherever the gradient is not round-off itself
        big = g0[n].abs() > 1e-5 * scale + 1e-4 * gmax
        assert float(((p0[n] - p1[n]).abs() * big).max()) <= 2e-5, n
        assert float((p0[n] - p1[n]).abs().max()) <= 2.01e-3, n
    assert n_with > 100


def test_sync_batchnorm_on_device_through_rccl(rccl_group):
    """SyncBatchNorm1d's packed all-gather (forward) and all-reduce (backward) on CUDA tensors through the one-rank
    group == nn.BatchNorm1d: output, input gradient, parameter gradients, running statistics."""
    parallel = rccl_group
    parallel.FORCE_EXCHANGE = True
    try:
        torch.manual_seed(3)
        x = (torch.randn(4096, 32, device="cuda") * 3 + 1).requires_grad_(True)
        x2 = x.detach().clone().requires_grad_(True)
        bn = torch.nn.BatchNorm1d(32, eps=1e-4, momentum=0.1).cuda()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.5, 0.5)
        sb = parallel.SyncBatchNorm1d(32, eps=1e-4, momentum=0.1).cuda()
        sb.load_state_dict(bn.state_dict())
        gy = torch.randn(4096, 32, device="cuda")
        y = bn(x)
        y.backward(gy)
        y2 = sb(x2)
        y2.backward(gy)
        torch.cuda.synchronize()
        assert torch.allclose(y, y2, rtol=1e-5, atol=1e-5)
        assert torch.allclose(x.grad, x2.grad, rtol=1e-4, atol=1e-5)
        assert torch.allclose(bn.weight.grad, sb.weight.grad, rtol=1e-4, atol=1e-3)
        assert torch.allclose(bn.bias.grad, sb.bias.grad, rtol=1e-4, atol=1e-3)
        assert torch.allclose(bn.running_mean, sb.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(bn.running_var, sb.running_var, rtol=1e-4, atol=1e-6)
        # an empty rank still takes part in the collective
        e = torch.zeros(0, 32, device="cuda", requires_grad=True)
        sb(e).sum().backward()
        torch.cuda.synchronize()
    finally:
        parallel.FORCE_EXCHANGE = False
