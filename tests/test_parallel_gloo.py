"""CPU, world_size 2 over gloo: the data-parallel plumbing.

  * BucketedGradReducer on a toy module: averaged gradients, parameters unused on every rank get grad None back,
    buckets start from the gradient hooks in index order, slowest-rank timing;
  * the REAL model (GeoFormer through the oracle's operators): two ranks with one scene each must end up with
    exactly the gradients a single process computes on the two-scene batch (BatchNorm in eval mode, dropout 0);
  * SyncBatchNorm1d: two ranks with half of the rows each == one process with all rows (output, input gradient,
    parameter gradients, running statistics).
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from geoformer_amd import parallel

    assert parallel.init_distributed("gloo") == world
    return parallel


def _toy_worker(rank, world, port, ret):
    parallel = _init(rank, world, port)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    unused = torch.nn.Linear(4, 4)  # never receives a gradient (decoder before prepare_epochs)
    half = torch.nn.Linear(3, 2)    # receives a gradient on rank 0 only (an empty-foreground batch on the other rank)
    mod = torch.nn.ModuleList([unused, net, half])  # (reverse registration order = bucket order: unused comes last)
    red = parallel.BucketedGradReducer(mod, bucket_bytes=64)  # tiny buckets: several collectives
    ok = len(red.ranges) >= 3
    for it in range(2):  # twice: prepare() must fully reset the state
        x = torch.full((6, 5), float(rank + 1 + it))
        red.prepare()
        y = net(x)
        loss = y.sum() + (half(y).sum() if rank == 0 else 0.0)
        loss.backward()
        # buckets leave in index order: rank 0 (every leading bucket complete) starts some inside its backward, rank 1
        # (no gradient for `half`, which sits in bucket 0) sends everything from finish() -- same sequence on both
        ok = ok and (red.launched_in_backward >= 1 if rank == 0 else red.launched_in_backward == 0)
        red.finish()
        # reference: plain autograd on fresh copies
        net2 = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
        net2.load_state_dict(net.state_dict())
        half2 = torch.nn.Linear(3, 2)
        half2.load_state_dict(half.state_dict())
        y2 = net2(x)
        (y2.sum() + (half2(y2).sum() if rank == 0 else 0.0)).backward()
        for p, p2 in zip(list(net.parameters()) + list(half.parameters()), list(net2.parameters()) + list(half2.parameters())):
            g = p2.grad if p2.grad is not None else torch.zeros_like(p2)
            outs = [torch.zeros_like(g) for _ in range(world)]
            dist.all_gather(outs, g)
            ok = ok and p.grad is not None and torch.allclose(p.grad, sum(outs) / world, atol=1e-6)
        ok = ok and all(p.grad is None for p in unused.parameters())  # untouched on every rank: optimizers skip it
    t = parallel.max_over_ranks(1.0 + rank)
    ok = ok and abs(t - float(world)) < 1e-9
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_grad_reducer_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_toy_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))


# ---- the real model -------------------------------------------------------------------------------------------
N_PTS = 1800


def _scenes():
    from geoformer_amd import scene

    out = []
    for seed in (31, 32):
        sc = scene.make_small_scene(2200, seed)
        out.append({k: (v[:N_PTS] if isinstance(v, np.ndarray) else v) for k, v in sc.items()})  # equal sizes: the
    return out                                                                                    # mean CE averages


def _model(batch_size):
    import argparse
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import train_dp

    args = argparse.Namespace(batch_size=batch_size, small=True, prepare_epochs=1, sync_bn=False, bn_eval=True)
    return train_dp.build(args, torch.device("cpu"))


def _grads(m):
    return {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m.named_parameters()}


def _model_worker(rank, world, port, ret):
    parallel = _init(rank, world, port)
    from geoformer_amd import scene
    from oracle import cpu_backend

    torch.set_num_threads(2)
    scenes = _scenes()
    with cpu_backend.installed():
        cfg, m, crit = _model(1)
        red = parallel.BucketedGradReducer(m, bucket_bytes=2 << 20)
        batch = scene.make_batch([scenes[rank]])
        # host RNG: the single-process run draws scene 0's permutation, then scene 1's, from one generator
        np.random.seed(3)
        n_fg = torch.zeros(1, dtype=torch.int64)
        if rank == 1:
            dist.recv(n_fg, src=0)
            np.random.choice(int(n_fg), min(int(n_fg), cfg.n_downsampling), replace=False)
        red.prepare()
        out = m(batch, 5)
        if rank == 0:
            dist.send(torch.tensor([m.last_sampling_indices.numel() if cfg.n_downsampling >= out["fg_idxs"].numel()
                                    else out["fg_idxs"].numel()], dtype=torch.int64), dst=1)
        loss, _ = crit(out, batch, 5)
        loss.backward()
        red.finish()
        ret[rank] = (float(loss), {k: (None if v is None else v.numpy()) for k, v in _grads(m).items()},
                     red.launched_in_backward, len(red.ranges))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process_on_the_concatenated_batch(oracle):
    from geoformer_amd import scene
    from oracle import cpu_backend

    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_model_worker, args=(world, port, ret), nprocs=world, join=True)
    with cpu_backend.installed():
        cfg, m, crit = _model(2)
        batch = scene.make_batch(_scenes())
        np.random.seed(3)
        m.zero_grad()
        out = m(batch, 5)
        loss, _ = crit(out, batch, 5)
        loss.backward()
        ref = _grads(m)
    l0, g0, early0, nb0 = ret[0]
    l1, g1, early1, nb1 = ret[1]
    assert abs((l0 + l1) / 2 - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    assert nb0 >= 3 and early0 >= 1 and early1 >= 1  # several buckets, some started inside the backward
    # relative to each parameter's own gradient scale -- except where the gradient is analytically zero and only
    # rounding noise is left (a bias in front of a soft-max: attn_mlp.2.bias, k_linear.bias): there relative to the
    # largest gradient of the model
    gmax = max(float(r.abs().max()) for r in ref.values() if r is not None)
    errs = []
    for name, r in ref.items():
        a, b = g0[name], g1[name]
        assert (a is None) == (r is None) and (b is None) == (r is None), name
        if r is None:
            continue
        assert np.array_equal(a, b), name  # both ranks hold the same averaged gradient
        scale = max(float(r.abs().max()), 1e-4 * gmax)
        errs.append((float(np.abs(a - r.numpy()).max()) / scale, name))
    assert max(errs)[0] < 2e-4, sorted(errs)[-6:]


# ---- SyncBatchNorm1d -----------------------------------------------------------------------------------------
def _bn_worker(rank, world, port, ret):
    parallel = _init(rank, world, port)
    rng = np.random.default_rng(0)
    x_all = torch.from_numpy(rng.standard_normal((64, 6)).astype(np.float32))
    w_all = torch.from_numpy(rng.standard_normal((64, 6)).astype(np.float32))
    bn = parallel.SyncBatchNorm1d(6, eps=1e-4, momentum=0.1)
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, 6))
        bn.bias.copy_(torch.linspace(-0.2, 0.3, 6))
    lo, hi = (0, 40) if rank == 0 else (40, 64)  # uneven split
    x = x_all[lo:hi].clone().requires_grad_()
    y = bn(x)
    (y * w_all[lo:hi]).sum().backward()
    for p in bn.parameters():
        dist.all_reduce(p.grad)  # what the gradient reducer does (sum; the mean's 1/world is a convention)
    ret[rank] = (y.detach().numpy(), x.grad.numpy(), bn.weight.grad.numpy(), bn.bias.grad.numpy(),
                 bn.running_mean.numpy(), bn.running_var.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sync_batchnorm_matches_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bn_worker, args=(world, port, ret), nprocs=world, join=True)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((64, 6)).astype(np.float32)).requires_grad_()
    w = torch.from_numpy(rng.standard_normal((64, 6)).astype(np.float32))
    bn = torch.nn.BatchNorm1d(6, eps=1e-4, momentum=0.1)
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, 6))
        bn.bias.copy_(torch.linspace(-0.2, 0.3, 6))
    y = bn(x)
    (y * w).sum().backward()
    y0, gx0, gw0, gb0, rm0, rv0 = ret[0]
    y1, gx1, gw1, gb1, rm1, rv1 = ret[1]
    assert np.abs(np.concatenate([y0, y1]) - y.detach().numpy()).max() < 1e-5
    assert np.abs(np.concatenate([gx0, gx1]) - x.grad.numpy()).max() < 1e-5
    assert np.abs(gw0 - bn.weight.grad.numpy()).max() < 1e-4 and np.abs(gb0 - bn.bias.grad.numpy()).max() < 1e-4
    assert np.abs(rm0 - bn.running_mean.numpy()).max() < 1e-6 and np.abs(rv0 - bn.running_var.numpy()).max() < 1e-5
    assert np.array_equal(rm0, rm1) and np.array_equal(rv0, rv1)


# ---- SyncBatchNorm + divergent ranks through the real model -----------------------------------------------------
def _syncbn_worker(rank, world, port, ret):
    """sync-bn on, rank 1's first batch has NO foreground: both ranks must issue the same sequence of collectives
    (the empty rank still takes part in every BatchNorm layer's exchange up to the point where ALL ranks leave the
    forward together), nobody hangs, and the next step -- both ranks with foreground -- trains normally."""
    parallel = _init(rank, world, port)
    import argparse
    import sys

    from geoformer_amd import scene
    from oracle import cpu_backend

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import train_dp

    torch.set_num_threads(2)
    counts = {"n": 0}
    for name in ("all_reduce", "all_gather_into_tensor"):
        orig = getattr(dist, name)

        def counted(*a, _orig=orig, **k):
            counts["n"] += 1
            return _orig(*a, **k)

        setattr(dist, name, counted)
    with cpu_backend.installed():
        args = argparse.Namespace(batch_size=1, small=True, prepare_epochs=1, sync_bn=True, bn_eval=False)
        cfg, m, crit = train_dp.build(args, torch.device("cpu"))
        assert m.rank_agreement is parallel.all_ranks_agree
        red = parallel.BucketedGradReducer(m, bucket_bytes=2 << 20)
        assert not red.overlap  # SyncBatchNorm layers present: buckets leave from finish() only
        batch = scene.make_batch([_scenes()[rank]])
        orig_bb = m.forward_backbone
        if rank == 1:  # an empty-foreground batch: every point predicted background
            def no_fg(*a, **k):
                feats, scores, preds = orig_bb(*a, **k)
                return feats, scores, torch.zeros_like(preds)

            m.forward_backbone = no_fg
        loss0, _ = train_dp.step(m, crit, red, None, batch, 5, 3)
        n0 = counts["n"]
        m.forward_backbone = orig_bb
        loss1, _ = train_dp.step(m, crit, red, None, batch, 5, 4)
        g = {n: (None if p.grad is None else p.grad.detach().clone().numpy()) for n, p in m.named_parameters()}
        ret[rank] = (loss0, n0, loss1, counts["n"] - n0, g)
    dist.barrier()
    dist.destroy_process_group()


def test_sync_bn_with_an_empty_foreground_rank_keeps_the_collective_sequence(oracle):
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_syncbn_worker, args=(world, port, ret), nprocs=world, join=True)
    (l00, n00, l01, n01, g0), (l10, n10, l11, n11, g1) = ret[0], ret[1]
    assert np.isnan(l00) and np.isnan(l10)  # BOTH ranks left the first forward (agreement), nobody computed a loss
    assert n00 == n10 and n01 == n11 and n01 > n00  # identical numbers of collectives on both ranks, in both steps
    assert np.isfinite(l01) and np.isfinite(l11)
    for name in g0:
        assert (g0[name] is None) == (g1[name] is None), name
        if g0[name] is not None:
            assert np.array_equal(g0[name], g1[name]), name  # the averaged gradient is the same tensor on both ranks


def test_sync_batchnorm_empty_rank_takes_part():
    """One rank with zero rows: it contributes count 0 and the other rank's statistics are those of its own rows."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bn_empty_worker, args=(world, port, ret), nprocs=world, join=True)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((40, 6)).astype(np.float32) * 3 + 100.0)  # |mean| >> std: no cancellation
    bn = torch.nn.BatchNorm1d(6, eps=1e-4, momentum=0.1)
    y = bn(x)
    assert np.abs(ret[0] - y.detach().numpy()).max() < 1e-4
    assert ret[1].shape == (0, 6)


def _bn_empty_worker(rank, world, port, ret):
    parallel = _init(rank, world, port)
    rng = np.random.default_rng(0)
    x_all = torch.from_numpy(rng.standard_normal((40, 6)).astype(np.float32) * 3 + 100.0)
    bn = parallel.SyncBatchNorm1d(6, eps=1e-4, momentum=0.1)
    x = (x_all if rank == 0 else x_all[:0]).clone().requires_grad_()
    y = bn(x)
    y.sum().backward()
    ret[rank] = y.detach().numpy()
    dist.barrier()
    dist.destroy_process_group()


# ---- bench.py --gpus N launches its own ranks -------------------------------------------------------------------
def test_bench_self_launch_prints_one_line_with_n_gpus():
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--plumbing-test"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3


# ---- tools/train_dp.run(): the loop bench.py times as secondary.train_step_b4 / train_dp_step -------------------------
def _run_worker(rank, world, port, ret):
    _init(rank, world, port)
    import sys

    from oracle import cpu_backend

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import train_dp

    torch.set_num_threads(2)
    with cpu_backend.installed():
        args = train_dp.default_args(steps=2, warmup=0, batch_size=1, points=2200, small=True, prepare_epochs=1, epoch=5,
                                     fg_frac=0.4, bucket_mb=2.0)
        res = train_dp.run(args, torch.device("cpu"))
    ret[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_train_dp_run_two_ranks(oracle):
    """The harness loop itself over gloo (the model on the host through the oracle's operators): foreground calibration
    broadcast from rank 0, two steps with the bucketed reducer, slowest-rank timing, the result dict bench.py prints."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run_worker, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0["n_gpus"] == 2 and r0["global_batch"] == 2 and r0["steps"] == 2
    assert r0["fg_bias_shift"] == r1["fg_bias_shift"] and r0["ms_per_step"] == r1["ms_per_step"]
    assert np.isfinite(r0["last_loss"]) and np.isfinite(r1["last_loss"]) and r0["backend"] == "gloo"
    assert len(r0["n_fg_per_step"]) == 2 and all(0 < n < 2200 for n in r0["n_fg_per_step"])
