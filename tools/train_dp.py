#!/usr/bin/env python3
"""Data-parallel training step of GeoFormer (BASELINE config 5: N ranks x `batch_size` scenes, RCCL gradient
all-reduce over xGMI for the training step only; the reference's intended launch is train.py:156-185).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        tools/train_dp.py --steps 10 --warmup 2 --batch-size 4 --points 150000 [--sync-bn] [--epoch 200]

One process per GPU (LOCAL_RANK selects the device; the reference's hard-coded device 0 wants
HIP_VISIBLE_DEVICES=<rank> instead when its own train.py is used).  Every rank builds its own synthetic scenes
(seed + rank) and rulebooks; the only exchange is BucketedGradReducer's bucketed all-reduce, started from gradient
hooks while the backward is still running (geoformer_amd/parallel.py), plus one packed all-reduce per BatchNorm layer
and direction with --sync-bn.  Timing: barrier + synchronize on both sides, max over ranks; rank 0 prints one JSON
line.  `run()` is the loop itself: bench.py calls it for its `secondary.train_step_b4` (one GPU) and
`secondary.train_dp_step` (N ranks) lines; tests/test_parallel_gloo.py drives it over gloo on the host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import geoformer_amd  # noqa: E402

geoformer_amd.configure_runtime()  # GPU_MAX_HW_QUEUES, before this process's first HIP call (a user's value wins)


def build(args, device):
    from geoformer_amd import parallel
    from geoformer_amd.model import GeoFormer, InstSetCriterion, load_config
    from tests.util import synthetic_state_dict

    over = dict(batch_size=args.batch_size, dec_dropout=0.0)
    if args.small:
        over.update(n_decode_point=128, n_query_points=16)
    if args.prepare_epochs is not None:
        over["prepare_epochs"] = args.prepare_epochs
    cfg = load_config("geoformer_scannet.yaml", **over)
    torch.manual_seed(0)  # identical initial weights on every rank
    m = GeoFormer(cfg)
    m.load_state_dict(synthetic_state_dict(m.state_dict(), 1))
    if args.sync_bn:
        parallel.convert_sync_batchnorm(m)
    m.to(device)
    m.train()
    if args.bn_eval:
        for mod in m.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.eval()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return cfg, m, InstSetCriterion(cfg)


def calibrate_foreground(m, batch, frac):
    """Random-init weights make the predicted-foreground share of a synthetic scene arbitrary (80-90 % with this seed);
    real ScanNet scenes have ~40 %.  Like bench.py's eval model, the background logits are shifted so
    that `frac` of the batch's points are predicted foreground -- from one no-grad forward of the model as it stands
    (training-mode BatchNorm: batch statistics).  Returns the shift (identical on every rank: same weights, rank 0's
    batch decides when the caller broadcasts it)."""
    with torch.no_grad():
        s = m(batch, 0)["semantic_scores"]
        margin = s[:, 4:].max(1)[0] - s[:, :4].max(1)[0]
        k = max(1, min(margin.numel(), int(round((1.0 - frac) * margin.numel()))))
        shift = float(torch.kthvalue(margin.float().flatten(), k)[0])
        m.semantic_linear.bias[:4] += shift
    return shift


def make_batches(args, rank, device, n):
    from geoformer_amd import scene

    out = []
    for i in range(n):
        scenes = []
        for b in range(args.batch_size):
            seed = args.seed + (rank * n + i) * args.batch_size + b
            scenes.append(scene.make_small_scene(args.points, seed) if args.small else scene.make_scene(args.points, seed))
        batch = scene.make_batch(scenes)
        out.append({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()})
    return out


def step(m, crit, red, opt, batch, epoch, np_seed):
    """One training step.  A rank whose batch has no foreground (mask_predictions None past prepare_epochs) skips its
    backward like train.py:68-69 does, but still takes part in the gradient exchange with zeros; with SyncBatchNorm the
    model itself makes all ranks leave the forward together (parallel.all_ranks_agree)."""
    np.random.seed(np_seed)
    red.prepare()  # zeroes the flat gradient buffer and points every p.grad into it (instead of zero_grad)
    out = m(batch, epoch)
    if epoch > m.prepare_epochs and out.get("mask_predictions") is None:
        red.finish()
        if opt is not None:
            opt.step()  # the other ranks' averaged gradients (none when every rank skipped): parameters stay in step
        return float("nan"), {}
    loss, info = crit(out, batch, epoch)
    if "n_fg_total" in out:
        info = dict(info, n_fg=out["n_fg_total"])
    loss.backward()  # buckets leave from the gradient hooks while this runs
    red.finish()
    if opt is not None:
        opt.step()
    return float(loss.detach()), info


def run(args, device, batches=None):
    """Build model / criterion / reducer / Adam, run `warmup` + `steps` training steps over two rotating batches and
    return the result dict (time = barrier + synchronise on both sides, max over ranks).  The process group, if any,
    is the caller's."""
    import torch.distributed as dist

    from geoformer_amd import parallel

    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    cfg, m, crit = build(args, device)
    red = parallel.BucketedGradReducer(m, bucket_bytes=int(args.bucket_mb * (1 << 20)),
                                       always_exchange=bool(getattr(args, "always_exchange", False)))
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3,
                           fused=device.type == "cuda")  # one launch per step on the GPU (the foreach form: ~5 ms of host time)
    nb = 2
    if batches is None:
        batches = make_batches(args, rank, device, nb)
    fg_shift = None
    if getattr(args, "fg_frac", None):
        fg_shift = calibrate_foreground(m, batches[0], args.fg_frac)
        if world > 1:  # every rank carries rank 0's value: the replicas must stay identical
            t = torch.tensor([fg_shift], dtype=torch.float64, device=device if device.type == "cuda" else "cpu")
            dist.broadcast(t, 0)
            with torch.no_grad():
                m.semantic_linear.bias[:4] += float(t.item()) - fg_shift
            fg_shift = float(t.item())
    sync = (lambda: torch.cuda.synchronize()) if device.type == "cuda" else (lambda: None)
    for i in range(args.warmup):
        step(m, crit, red, opt, batches[i % len(batches)], args.epoch, 100 * rank + i)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    from geoformer_amd import _lib as _gl

    wait_ns = _gl.load().gf_dev_host_wait_ns if device.type == "cuda" else (lambda reset: 0)
    _gl.host_wait_s[0] = 0.0
    wait_ns(1)
    t0 = time.perf_counter()
    early = 0
    loss = float("nan")
    n_fg = []
    for i in range(args.steps):
        loss, info = step(m, crit, red, opt, batches[i % len(batches)], args.epoch, 100 * rank + args.warmup + i)
        early += red.launched_in_backward
        if "n_fg" in info:
            n_fg.append(info["n_fg"])
    # this rank's host share of a step: the loop's issue time minus the package's own blocking waits (N ranks share one
    # host's cores: the first multi-GPU run reads here whether the Python side is what limits scaling)
    host_busy = time.perf_counter() - t0 - _gl.host_wait_s[0] - wait_ns(0) * 1e-9
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, device if device.type == "cuda" else None)
    busy = [host_busy]
    if world > 1:
        hb = torch.zeros(world, dtype=torch.float64, device=device if device.type == "cuda" else "cpu")
        hb[rank] = host_busy
        dist.all_reduce(hb)
        busy = [float(x) for x in hb.tolist()]
    host_ms = None
    if getattr(args, "host_split", 0):
        # untimed extra steps: when the host has queued a whole step against when the device has finished it
        th = 0.0
        for i in range(args.host_split):
            sync()
            h0 = time.perf_counter()
            step(m, crit, red, opt, batches[i % len(batches)], args.epoch, 7000 + i)
            th += time.perf_counter() - h0
        sync()
        host_ms = round(th / args.host_split * 1e3, 2)
    scenes = world * args.batch_size * args.steps
    return {"metric": "training scenes/sec (fwd + criterion + bwd + all-reduce + Adam)", "value": round(scenes / dt, 3),
            "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "global_batch": world * args.batch_size,
            "points_per_scene": args.points, "points_per_batch": [int(b["locs"].shape[0]) for b in batches],
            "epoch": args.epoch, "prepare_epochs": int(m.prepare_epochs), "sync_bn": bool(args.sync_bn),
            "grad_floats": red.numel(), "buckets": len(red.ranges),
            "buckets_started_inside_backward_per_step": round(early / max(args.steps, 1), 2),
            "fg_frac_target": getattr(args, "fg_frac", None), "fg_bias_shift": fg_shift,
            "host_ms_per_step": host_ms,
            "host_busy_ms_per_step_per_rank": [round(x / max(args.steps, 1) * 1e3, 2) for x in busy],
            "threads_per_rank": torch.get_num_threads(), "n_fg_per_step": n_fg, "last_loss": loss, "backend": dist.get_backend() if world > 1 else "none", "data": "synthetic"}


def default_args(**over):
    """The argument namespace of `main` with its defaults (for callers of `run`)."""
    return parser().parse_args([], argparse.Namespace(**over))


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-size", type=int, default=4)
    ap.add_argument("--points", type=int, default=150_000)
    ap.add_argument("--epoch", type=int, default=200)
    ap.add_argument("--prepare-epochs", type=int, default=None)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--bucket-mb", type=float, default=8.0)
    ap.add_argument("--sync-bn", action="store_true")
    ap.add_argument("--bn-eval", action="store_true")
    ap.add_argument("--fg-frac", type=float, default=None,
                    help="shift the background logits so that this share of the first batch's points is predicted foreground")
    ap.add_argument("--host-split", type=int, default=0,
                    help="extra untimed steps that measure when the host has queued a step (host_ms_per_step)")
    ap.add_argument("--small", action="store_true", help="small scenes / heads (host-side test runs)")
    return ap


def main(argv=None):
    args = parser().parse_args(argv)
    import torch.distributed as dist

    from geoformer_amd import parallel

    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("tools/train_dp.py needs a GPU: the HIP operators have no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    world = parallel.init_distributed("nccl")
    if world > 1:  # the ranks share one host: every rank's framework threads get their share of its cores
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // world))
    res = run(args, device)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
