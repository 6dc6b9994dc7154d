#!/usr/bin/env python3
"""Headline benchmark: scenes/sec of the GeoFormer eval forward on ScanNet-like scenes.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one full ``GeoFormer.forward(batch, epoch, training=False)`` over one ~150k-point synthetic
scene (BASELINE.json configs[1]; test yaml: nq=256, nc=2048, batch 1) with the batch dict already
resident in HBM: voxel mean, 13 rulebooks, 71 sparse convs, semantic head, FPS, ball query, grouping,
kNN graph + geodesic BFS, 4 decoder layers, dynamic-conv mask head, proposals.  The loop is a serving loop: the
forward of scene i returns once everything up to the copy of the accepted-proposal count is queued, and the proposals
of scene i are collected right after scene i+1 has been issued (``defer_proposals``); all K scenes, proposals
included, are complete inside the timed region.  The steps rotate over
``--scenes`` (8) different resident scenes (seeds 1234, 1235, ...), so no step finds the previous step's
tables or features in L2/MALL.  Weights are random-init of the real architecture (no checkpoints
offline); the semantic head's bias is shifted so ~40 % of the points are foreground like a real scene
(SURVEY.md App. B #22).  Scenes are independent, so N ranks run N replicas (no data-path collective;
scaling "weak").

The JSON line also carries
  roofline       -- the level-1 (C=16) gather-MFMA sparse-conv launches, HBM-bound: algorithmic bytes
                    4*(R*Cin + M*Cout + K*Cin*Cout) + 8*R per launch / mean launch time from HIP events
                    recorded natively around those launches inside the timed region;
  roofline_convs -- all 71 sparse convolutions of a forward: sum of their algorithmic bytes / sum of their
                    spans (events around every conv call, separate untimed pass);
  cpu_baseline   -- the same forward of scene 0 through the build's model on the host cores with the oracle's
                    C operators ("port"; rank 0, N=1 only), with per-stage seconds;
  parity_s150k   -- max-abs differences between that host forward and the GPU forward of the same scene under
                    the same numpy seed (the pytest suite holds the asserting version: tests/test_gpu_fullsize.py);
  secondary      -- the nq=128 train-yaml variant of config 2 (config/geoformer_scannet.yaml), a few steps.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROBE_EVERY = 4  # timed steps between two probed ones
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# HBM-side bytes per launch of the roofline kernel from rocprofv3 PMC passes on scene 1234 (separate --pmc passes,
# FETCH_SIZE + WRITE_SIZE in KiB; file and command in profiles/README.md).  Far below the algorithmic figure: the
# ~6 re-reads of every input row are served by L1/L2.
PMC_TRAFFIC = {"bytes": (14998 + 8882) * 1024, "source": "profiles/r1_e_pmc_conv_l1.md"}
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_conv_l1_latest.json")
if os.path.exists(PMC_FILE):
    PMC_TRAFFIC = json.load(open(PMC_FILE))


def build_model(device, nfg_frac=0.4, probe_batch=None, bias_shift=None, cfg_name="test_geoformer_scannet.yaml"):
    """The benchmark model.  probe_batch: derive the background-logit shift that makes ~nfg_frac of the points
    foreground from one forward; bias_shift: apply a shift derived elsewhere (the host model of the parity /
    cpu_baseline leg must carry exactly the GPU model's value)."""
    from geoformer_amd.model import GeoFormer, load_config
    from tests.util import synthetic_state_dict

    cfg = load_config(cfg_name)
    m = GeoFormer(cfg)
    m.load_state_dict(synthetic_state_dict(m.state_dict(), 0))
    m.to(device)
    m.eval()
    if bias_shift is None and probe_batch is not None:
        with torch.no_grad():
            s = m(probe_batch, 0, training=False)["semantic_scores"]
            margin = s[:, 4:].max(1)[0] - s[:, :4].max(1)[0]
            bias_shift = float(torch.quantile(margin.float().cpu(), 1.0 - nfg_frac))
    if bias_shift is not None:
        with torch.no_grad():
            # in place on the parameter itself (bumps its version counter: the fused inference paths cache folded
            # copies of the parameters and re-derive them when a version changes; writes through `.data` are invisible)
            m.semantic_linear.bias[:4] += bias_shift
    m._bench_bias_shift = bias_shift
    return m


def to_device(batch, device):
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}


def _conv_bytes(R, M, K, Cin, Cout, residual=False):
    """every rule gathers one input row, every output row is written once, weights once,
    two int32 per rule; the residual epilogue reads one more [M, Cout] operand."""
    return 4 * (R * Cin + M * Cout + K * Cin * Cout) + 8 * R + (4 * M * Cout if residual else 0)


def _read_probe(max_records=8192):
    """Records of the native conv probe (include/geoformer_hip_dev.h: gf_dev_unet_probe_read2): list of
    (level, kind, K, Cin, Cout, M_in, M_out, residual, rules, microseconds between events recorded around the launch,
    microseconds between the events bound to the kernel itself or -1)."""
    import ctypes

    from geoformer_amd import _lib

    meta = (ctypes.c_int * (9 * max_records))()
    us = (ctypes.c_float * max_records)()
    usk = (ctypes.c_float * max_records)()
    n = _lib.load().gf_dev_unet_probe_read2(max_records, ctypes.cast(meta, ctypes.c_void_p), ctypes.cast(us, ctypes.c_void_p),
                                            ctypes.cast(usk, ctypes.c_void_p))
    if n < 0:
        raise RuntimeError("gf_dev_unet_probe_read2 failed")
    return [tuple(meta[9 * i:9 * i + 9]) + (float(us[i]), float(usk[i])) for i in range(n)]


class ConvProbe:
    """HIP events around the level-1 16->16 submanifold conv launches of the residual blocks, recorded by the native
    U-Net executor on the stream the kernels run on, inside the timed region (every PROBE_EVERY-th step)."""

    def __init__(self, batches):
        from geoformer_amd import _lib, sparse

        self.lib = _lib.load()
        self.R = {}
        for b in batches:  # level-1 rule counts of the benchmark scenes (setup, untimed)
            coords = b["voxel_locs"].int().contiguous()
            shape = tuple(int(x) for x in b["spatial_shape"])
            M = coords.shape[0]
            rules = sparse.subm_rules(coords, sparse.build_index(coords, 1, shape))
            self.R[M] = int((rules.nbr[:, :M] >= 0).sum().item())
        self.recs = []

    def arm(self, on):
        # probed steps alternate between events bound to the kernel launch (mode 3) and events recorded before / after
        # the launch (mode 1): binding events changes what surrounds the launch, so the two are not taken together
        if on:
            self.nprobed = getattr(self, "nprobed", 0) + 1
        self.lib.gf_dev_unet_probe((3 if self.nprobed % 2 else 1) if on else 0)

    def close(self):
        self.lib.gf_dev_unet_probe(0)
        self.recs += _read_probe()

    def result(self):
        if not self.recs:
            return None
        # the launch's duration: the events bound to the kernel itself (the dispatch's begin / end timestamps, what the
        # rocprofv3 kernel trace under profiles/ reports for the same kernel); the events recorded before / after the
        # launch on its stream additionally hold the command processor's handling of the event packets and are kept
        # beside it ("us_per_launch_bracketed")
        brack = [r[9] for r in self.recs if r[10] <= 0]
        recs = [r for r in self.recs if r[10] > 0]
        bound = bool(recs)
        if not bound:
            recs = self.recs
        us = [r[10] if bound else r[9] for r in recs]
        byt = [_conv_bytes(self.R[r[6]], r[6], 27, 16, 16, bool(r[7])) for r in recs]
        # mean of the per-launch rates weighted by time = total bytes / total time
        ach = sum(byt) / (sum(us) * 1e-6) / 1e9
        Ms = sorted({r[6] for r in recs})
        return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": PMC_TRAFFIC["bytes"],
                "traffic_source": PMC_TRAFFIC["source"],
                "kernel": "level-1 subm 3x3x3 16->16 launches of the residual blocks (first conv: BN+ReLU prologue and "
                          "BN+ReLU epilogue; second conv: residual epilogue), k_conv_g16p",
                "launches": len(us), "us_per_launch": round(float(np.mean(us)), 2),
                "us_per_launch_bracketed": round(float(np.mean(brack)), 2) if brack else None,
                "launches_bracketed": len(brack),
                "algorithmic_bytes": int(np.mean(byt)), "rules": {M: self.R[M] for M in Ms}, "voxels": Ms,
                "sampling": f"every such launch of every {PROBE_EVERY}th timed step, inside the timed region, on the "
                            "stream the kernel runs on; " +
                            ("start / stop events bound to the kernel launch (hipExtLaunchKernelGGL)" if bound else
                             "events recorded before / after the launch")}


def all_convs_roofline(model, batches, reps=2):
    """Every sparse-conv launch of a forward between two events (untimed extra passes through the native executor's
    probe): sum of algorithmic bytes and flops over the 71 convolutions / sum of their launch durations."""
    from geoformer_amd import _lib

    lib = _lib.load()
    lib.gf_dev_unet_probe(2)
    try:
        for b in batches[:reps]:
            with torch.no_grad():
                model(b, 0, training=False)  # backbone + semantic head only (epoch <= prepare_epochs)
        torch.cuda.synchronize()
    finally:
        lib.gf_dev_unet_probe(0)
    recs = _read_probe()
    us = sum(r[9] for r in recs)
    byt = fl = 0
    per_level = {}
    for level, kind, K, Cin, Cout, M_in, M_out, res, R, t, _tk in recs:
        R = M_out if R < 0 else R  # 1x1x1 convs: one rule per row
        bb = _conv_bytes(R, M_out, K, Cin, Cout, bool(res))
        byt += bb
        fl += 2 * R * Cin * Cout
        pl = per_level.setdefault(level + 1, [0, 0.0, 0])
        pl[0] += bb; pl[1] += t; pl[2] += 1
    nf = len(batches[:reps])
    return {"bound": "hbm", "achieved": round(byt / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(byt / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "convs_per_forward": len(recs) // nf,
            "us_per_forward": round(us / nf, 1), "algorithmic_bytes_per_forward": byt // nf,
            "gflop_per_forward": round(fl / nf / 1e9, 2), "tflops": round(fl / (us * 1e-6) / 1e12, 2),
            "by_level": {str(l): {"convs": v[2] // nf, "us": round(v[1] / nf, 1), "GB/s": round(v[0] / (v[1] * 1e-6) / 1e9, 1)}
                         for l, v in sorted(per_level.items())},
            "note": "events around every conv launch of the native U-Net executor (separate untimed passes)"}


class StageTimer:
    """Wall-clock seconds per stage of the host forward (backbone / aggregator / kNN / BFS / decoder /
    mask head)."""

    def __init__(self, model):
        import geoformer_amd.model.geoformer as G

        self.t = {}
        self.G, self.orig_knn = G, G.knn_graphs
        self._wrap_attr(model, "forward_backbone", "backbone")
        self._wrap_attr(model, "forward_aggregator", "aggregator")
        self._wrap_attr(model, "forward_decoder", "decoder")
        self._wrap_attr(model, "get_mask_prediction", "mask_head")
        self._wrap_attr(model, "generate_proposal", "proposals")
        G.knn_graphs = self._timed(G.knn_graphs, "knn")
        self.orig_geo = G.cal_geodesic
        G.cal_geodesic = self._timed(G.cal_geodesic, "knn+bfs")

    def _timed(self, fn, name):
        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            self.t[name] = self.t.get(name, 0.0) + time.perf_counter() - t0
            return r

        return w

    def _wrap_attr(self, model, attr, name):
        setattr(model, attr, self._timed(getattr(model, attr), name))

    def close(self):
        self.G.knn_graphs, self.G.cal_geodesic = self.orig_knn, self.orig_geo

    def stages(self):
        t = dict(self.t)
        if "knn+bfs" in t:
            t["bfs"] = t.pop("knn+bfs") - t.get("knn", 0.0)
        if "decoder" in t:  # relative_position_embedding (inside forward_decoder) is part of the decoder stage
            pass
        return {k: round(v, 3) for k, v in t.items()}


def cpu_forward(batch, bias_shift, seed, threads):
    """One eval forward of `batch` through the build's model on the host with the oracle's C operators."""
    from oracle import cpu_backend
    from oracle import oracle as orc

    torch.set_num_threads(threads)
    L = orc.lib()
    L.orc_set_threads.restype = int
    omp = int(L.orc_set_threads(threads))
    with cpu_backend.installed(), torch.no_grad():
        m = build_model("cpu", bias_shift=bias_shift)
        st = StageTimer(m)
        try:
            np.random.seed(seed)
            t = time.perf_counter()
            out = m(batch, 300, training=False)
            dt = time.perf_counter() - t
        finally:
            st.close()
    return out, dt, st.stages(), omp


def cpu_baseline_and_parity(model, batch, dev, seed=4321):
    """cpu_baseline (the host forward of scene 0, all host cores) and parity_s150k (its outputs against the GPU
    forward of the same scene under the same numpy seed)."""
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, int(os.environ.get("GF_CPU_BASELINE_THREADS", "64"))))
    host_batch = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in batch.items()}
    outc, dt, stages, omp = cpu_forward(host_batch, model._bench_bias_shift, seed, threads)
    n = int(host_batch["locs"].shape[0])
    base = {"value": round(1.0 / dt, 5), "unit": "scenes/s", "cores": omp, "kind": "port",
            "sample": f"one eval forward of benchmark scene 0 ({n} points, N_fg={int(outc['fg_idxs'].shape[0])}); "
                      f"oracle C operators on {omp} OpenMP threads, torch modules on {torch.get_num_threads()} "
                      f"threads ({cores} host cores); {dt:.1f} s",
            "seconds": round(dt, 2), "stage_seconds": stages}
    np.random.seed(seed)
    with torch.no_grad():
        outg = model(batch, 300, training=False)
    torch.cuda.synchronize()
    par = {"semantic_scores_maxabs": float((outg["semantic_scores"].cpu() - outc["semantic_scores"]).abs().max()),
           "semantic_scores_scale": float(outc["semantic_scores"].abs().max())}
    fg_g, fg_c = outg["fg_idxs"].cpu().numpy(), outc["fg_idxs"].numpy()
    par["fg_idxs_xor"] = int(np.setxor1d(fg_g, fg_c).size)
    if par["fg_idxs_xor"] == 0:
        mg, mc = outg["mask_predictions"][-1], outc["mask_predictions"][-1]
        par["cls_logits_maxabs"] = float((mg["cls_logits"].cpu() - mc["cls_logits"]).abs().max())
        par["mask_logits_maxabs"] = float((mg["mask_logits"][0].cpu() - mc["mask_logits"][0]).abs().max())
        par["mask_logits_scale"] = float(mc["mask_logits"][0].abs().max())
        pg, pc = outg["proposal_scores"], outc["proposal_scores"]
        par["proposals"] = [len(pg[0]), len(pc[0])]
        if len(pg[0]) == len(pc[0]) and len(pg[0]):
            par["proposal_scores_maxabs"] = float((pg[1].cpu() - pc[1]).abs().max())
    else:
        par["note"] = ("class decisions differ on near-tie points, downstream point sets are not comparable one to one; "
                       "tests/test_gpu_fullsize.py compares the stages on identical foreground sets")
    par["tolerance"] = "1e-4 abs, or 32 fp32 ulps of the tensor's largest magnitude (*_scale) where that is more"
    return base, par


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--points", type=int, default=150_000)
    ap.add_argument("--scenes", type=int, default=8, help="resident scenes the steps rotate over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP operators have no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from geoformer_amd import scene

    if os.environ.get("GF_CONV_CHUNKS"):  # dev knob: waves of the level-1 conv kernel (include/geoformer_hip_dev.h)
        from geoformer_amd import sparse

        sparse.dev_conv_chunks(int(os.environ["GF_CONV_CHUNKS"]))
    # every rank gets its own scenes: replicas of the same workload
    ns = max(1, args.scenes)
    batches = [to_device(scene.make_batch([scene.make_scene(args.points, 1234 + rank * ns + i)]), dev)
               for i in range(ns)]
    model = build_model(dev, probe_batch=batches[0])
    Ms = [int(b["voxel_locs"].shape[0]) for b in batches]
    probe = ConvProbe(batches)

    class Loop:
        """One scene per step.  The forward queues everything up to the copy of the accepted-proposal count and
        returns (defer_proposals); the proposals of scene i are collected -- count read, membership scatter queued --
        right after scene i+1 has been issued, when that count has long reached the host, so the loop never sits in
        the forward's last read-back.  Every scene's outputs, proposals included, are complete when `finish` returns."""

        def __init__(self):
            self.prev = None

        def step(self, i, m=model):
            np.random.seed(1000 + i)
            with torch.no_grad():
                out = m(batches[i % ns], 300, training=False, defer_proposals=True)
            self.finish()
            self.prev = out
            return out

        def finish(self):
            if self.prev is not None and not isinstance(self.prev.get("proposal_scores"), (tuple, type(None))):
                self.prev["proposal_scores"] = self.prev["proposal_scores"].get()
            self.prev = None

    loop = Loop()
    step = loop.step
    for i in range(max(args.warmup, 0)):
        out = step(i)
    loop.finish()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # the probe's event pairs go around the level-1 conv launches of every PROBE_EVERY-th timed step (two event
    # records per launch, issued by the native executor itself)
    t0 = time.perf_counter()
    for i in range(args.steps):
        probe.arm(i % PROBE_EVERY == 0)
        out = step(args.warmup + i)
    loop.finish()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    probe.close()
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        n_fg = int(out["fg_idxs"].shape[0])
        res = {
            "metric": "scenes/sec forward (ScanNetV2 ~150k pts)",
            "value": round(world * args.steps / elapsed, 3),
            "unit": "scenes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"S150k eval forward, batch=1 per GPU, rotating over {ns} resident scenes, "
                                   "config/test_geoformer_scannet.yaml (nq=256, nc=2048, 4 decoder layers), "
                                   "random-init weights; proposals of scene i collected after scene i+1 is issued",
                       "points": [int(b["locs"].shape[0]) for b in batches], "voxels": Ms, "n_fg_last": n_fg,
                       "parallelism": f"replicas x{world}"},
            "roofline": probe.result(),
        }
        res["roofline_convs"] = all_convs_roofline(model, batches)
        if not args.no_secondary:
            m128 = build_model(dev, bias_shift=model._bench_bias_shift, cfg_name="geoformer_scannet.yaml")
            k = min(args.steps, 16)
            for i in range(max(min(args.warmup, 4), ns)):  # every scene once: one-time costs per (model, scene) stay out
                step(i, m128)
            loop.finish()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(k):
                step(100 + i, m128)
            loop.finish()
            torch.cuda.synchronize()
            e1 = time.perf_counter() - t1
            res["secondary"] = {"nq128_train_yaml_eval_forward": {
                "value": round(k / e1, 3), "unit": "scenes/s", "ms_per_step": round(e1 / k * 1e3, 3), "steps": k,
                "config": "config/geoformer_scannet.yaml (nq=128), same scenes, one GPU"}}
            del m128
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"], res["parity_s150k"] = cpu_baseline_and_parity(model, batches[0], dev)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
