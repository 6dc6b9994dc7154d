"""CPU, world_size 2 over gloo: the data-parallel plumbing (flat gradient all-reduce with unused
parameters, slowest-rank timing)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from geoformer_amd import parallel

    assert parallel.init_distributed("gloo") == world
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    unused = torch.nn.Linear(4, 4)  # never receives a gradient (decoder before prepare_epochs)
    mod = torch.nn.ModuleList([net, unused])
    x = torch.full((6, 5), float(rank + 1))
    net(x).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    red = parallel.FlatGradAllReduce(mod, bucket_bytes=64)  # tiny buckets: several collectives
    red.reduce()
    gathered = [[torch.zeros_like(g) for _ in range(world)] for g in local]
    for g, out in zip(local, gathered):
        dist.all_gather(out, g)
    ok = all(torch.allclose(p.grad, sum(out) / world, atol=1e-6) for p, out in zip(net.parameters(), gathered))
    ok = ok and all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in unused.parameters())
    t = parallel.max_over_ranks(1.0 + rank)
    ok = ok and abs(t - float(world)) < 1e-9
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))
