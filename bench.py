#!/usr/bin/env python3
"""Headline benchmark: scenes/sec of the GeoFormer eval forward on ScanNet-like scenes.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one full ``GeoFormer.forward(batch, epoch, training=False)`` over one ~150k-point synthetic
scene (BASELINE.json configs[1]; test yaml: nq=256, nc=2048, batch 1) with the batch dict already
resident in HBM: voxel mean, 13 rulebooks, 71 sparse convs, semantic head, FPS, ball query, grouping,
kNN graph + geodesic BFS, 4 decoder layers, dynamic-conv mask head, proposals.  The loop is the reference's test loop
(test.py:60-110): ONE scene at a time on the current stream (the forward of scene i returns once everything up to the
copy of the accepted-proposal count is queued -- ``defer_proposals`` -- and its proposals are collected after scene i+1
has been issued).  ``--staggered``: the serving loop of geoformer_amd/serving.py instead (two scenes in flight on two
streams: scene i's decoder + mask head under scene i+1's sampling / BFS stretch); it is timed as
``secondary.staggered_two_in_flight`` in the default run -- it wins on some boxes and loses on others, so it
is not the headline.  Either way all K scenes, proposals included, are complete inside the timed region.  The steps rotate over
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
  parity_s150k   -- max-abs differences between that host forward and the GPU forward of the same scene under
                    the same numpy seed (the pytest suite holds the asserting version: tests/test_gpu_fullsize.py), with
                    both bounds evaluated (1e-4 absolute; 64 fp32 epsilons of the tensor's magnitude);
  parity_s150k_calibrated -- the same under weights with trained-net-like activation scales, 1e-4 ABSOLUTE only;
  roofline_decoder / roofline_mask_head / roofline_bfs / sampling -- the other blocks north_star names, from event
                    pairs around their launches in untimed extra forwards (OpProbe): cross-attention against the fp32
                    MFMA peak, mask head against both peaks, the BFS's achieved GB/s, microseconds per sampling pick;
  cpu_baseline   -- the same forward of scene 0 through the build's model on the host cores with the oracle's
                    C operators ("port"; rank 0, N=1 only), with per-stage seconds;
  secondary      -- nq128_train_yaml_eval_forward (config 2's other yaml), staggered_two_in_flight (the headline workload
                    through the serving loop), fresh_scenes (24 timed steps, every one a never-before-seen scene of a
                    new size: nothing cached per scene size can help), train_step_b4 (config 3: batch 4, ~550k
                    points, forward + criterion + backward + Adam, both epoch regimes), fs_1shot / fs_5shot (config 4:
                    S150k query + k full support scenes), fs_train_episode_b4 (its training-mode episode) and, for N > 1, train_dp_step (config 5: every rank a batch of 4,
                    bucketed RCCL gradient all-reduce; tools/train_dp.py's loop).

`--gpus N` without a torchrun environment launches the N ranks itself (python -m torch.distributed.run ... bench.py,
as a child process, before this process touches a GPU) and relays rank 0's JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import geoformer_amd  # noqa: E402

geoformer_amd.configure_runtime()  # GPU_MAX_HW_QUEUES, before this process's first HIP call (a user's value wins)

PROBE_EVERY = 4  # timed steps between two probed ones
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_16x16x4_f32)
# the parity bound of the whole S150k forward: 1e-4 absolute (north_star), or 64 fp32 ulps of the
# tensor's largest magnitude where that is more -- the float64 arbiter test (tests/test_gpu_fullsize.py) shows the GPU
# and the fp32 host forward each this close to the float64 result on activations of magnitude ~60
PARITY_ULPS = 64
# HBM-side bytes per launch of the roofline kernel from rocprofv3 PMC passes on scene 1234 (separate --pmc passes,
# FETCH_SIZE + WRITE_SIZE in KiB; file and command in profiles/README.md).  Far below the algorithmic figure: the
# ~6 re-reads of every input row are served by L1/L2.
PMC_TRAFFIC = {"bytes": (14998 + 8882) * 1024, "source": "profiles/r1_e_pmc_conv_l1.md"}
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_conv_l1_latest.json")
if os.path.exists(PMC_FILE):
    PMC_TRAFFIC = json.load(open(PMC_FILE))
# a counter figure is quoted only for the kernel source it was measured on: the collecting script stores the hash of
# csrc/spconv_conv.hip (file times mean nothing in a fresh clone); after an edit of that file the line says so instead
_CONV_SRC = os.path.join(ROOT, "geoformer_amd", "csrc", "spconv_conv.hip")
if os.path.exists(_CONV_SRC):
    import hashlib

    _sha = hashlib.sha256(open(_CONV_SRC, "rb").read()).hexdigest()
    if PMC_TRAFFIC.get("kernel_source_sha256") != _sha:
        PMC_TRAFFIC = {"bytes": None, "source": "stale: " + str(PMC_TRAFFIC.get("source")) + " was collected on another "
                       "version of csrc/spconv_conv.hip -- re-run tools/pmc_conv.sh"}
# the same for the whole conv family of a forward (tools/pmc_conv_family.sh): quoted while the hash of the three sources it
# covers matches
PMC_FAMILY = {"bytes_per_forward": None, "source": "no profiles/pmc_conv_family_latest.json"}
_PMC_FAMILY_FILE = os.path.join(ROOT, "profiles", "pmc_conv_family_latest.json")
if os.path.exists(_PMC_FAMILY_FILE):
    import hashlib

    PMC_FAMILY = json.load(open(_PMC_FAMILY_FILE))
    _srcs = [os.path.join(ROOT, "geoformer_amd", "csrc", f) for f in ("spconv_conv.hip", "spconv_lw.hip", "unet_exec.hip")]
    if all(os.path.exists(f) for f in _srcs):
        _sha = hashlib.sha256(b"".join(open(f, "rb").read() for f in _srcs)).hexdigest()
        if PMC_FAMILY.get("kernel_source_sha256") != _sha:
            PMC_FAMILY = {"bytes_per_forward": None, "source": "stale: " + str(PMC_FAMILY.get("source"))[:120] + " ... was "
                          "collected on another version of the conv sources -- re-run tools/pmc_conv_family.sh"}
# average duration of the same launches in the committed rocprofv3 kernel trace of `bench.py` (tools/bench_trace.sh
# writes it): {"us_per_launch": ..., "source": ...}
ROCPROF_FILE = os.path.join(ROOT, "profiles", "rocprof_conv_l1_latest.json")
ROCPROF_FAMILY_FILE = os.path.join(ROOT, "profiles", "rocprof_conv_family_latest.json")


def _family_rocprof(fam):
    """The committed rocprofv3 kernel trace's figure for the same 71 launches (summed KERNEL durations per forward) beside
    the live one (events around every launch: they also see the dependent-dispatch gap in front of each launch, ~3 us x 71).
    `frac` stays on the slower, live figure."""
    if not os.path.exists(ROCPROF_FAMILY_FILE):
        return {}
    rp = json.load(open(ROCPROF_FAMILY_FILE))
    us = float(rp["us_per_forward"])
    return {"frac_by_launch_events": fam["frac"], "rocprof_us_per_forward": us,
            "frac_by_rocprof_trace": round(fam["algorithmic_bytes_per_forward"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "rocprof_source": rp.get("source"),
            "frac_note": "frac = the slower of the two: events recorded around every launch inside bench.py (they include the "
                         "dependent-dispatch gap in front of a launch) against the kernels' own durations in the committed trace"}


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
        # the rocprofv3 kernel trace of this command (profiles/, tools/bench_trace.sh) is the clock the line is checked
        # against, and it reads this kernel ~7 % longer than the launch-bound events do (and ~7 % shorter than the
        # bracketing events): when the committed trace summary is there, `achieved` / `frac` are quoted on the SLOWER
        # of the two clocks and both are kept beside it
        ach_events = ach
        prof = None
        if os.path.exists(ROCPROF_FILE):
            prof = json.load(open(ROCPROF_FILE))
            # the trace's average covers every such launch of every scene of the rotation: bytes averaged the same way
            # (a forward issues seven of these launches, four of them with the residual epilogue)
            byt_all = float(np.mean([(3 * _conv_bytes(R, M, 27, 16, 16, False) + 4 * _conv_bytes(R, M, 27, 16, 16, True)) / 7.0
                                     for M, R in self.R.items()]))
            ach_prof = byt_all / (prof["us_per_launch"] * 1e-6) / 1e9
            ach = min(ach, ach_prof)
        return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4),
                "frac_by_launch_events": round(ach_events / HBM_PEAK_GBS, 4),
                "frac_by_rocprof_trace": round(ach_prof / HBM_PEAK_GBS, 4) if prof else None,
                "rocprof_us_per_launch": prof["us_per_launch"] if prof else None,
                "rocprof_algorithmic_bytes": int(byt_all) if prof else None,
                "rocprof_source": prof["source"] if prof else None,
                "traffic": PMC_TRAFFIC["bytes"],
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
    byt = fl = comp = 0
    per_level = {}
    for level, kind, K, Cin, Cout, M_in, M_out, res, R, t, _tk in recs:
        R = M_out if R < 0 else R  # 1x1x1 convs: one rule per row
        bb = _conv_bytes(R, M_out, K, Cin, Cout, bool(res))
        byt += bb
        # SURVEY.md 8(d)'s compulsory lower bound: every input element read once, every output element written once
        comp += 4 * (M_in * Cin + M_out * Cout)
        fl += 2 * R * Cin * Cout
        pl = per_level.setdefault(level + 1, [0, 0.0, 0])
        pl[0] += bb; pl[1] += t; pl[2] += 1
    nf = len(batches[:reps])
    return {"bound": "hbm", "achieved": round(byt / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(byt / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "convs_per_forward": len(recs) // nf,
            "us_per_forward": round(us / nf, 1), "algorithmic_bytes_per_forward": byt // nf,
            "compulsory_bytes_per_forward": comp // nf,
            "compulsory_frac": round(comp / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "gflop_per_forward": round(fl / nf / 1e9, 2), "tflops": round(fl / (us * 1e-6) / 1e12, 2),
            "by_level": {str(l): {"convs": v[2] // nf, "us": round(v[1] / nf, 1), "GB/s": round(v[0] / (v[1] * 1e-6) / 1e9, 1)}
                         for l, v in sorted(per_level.items())},
            "note": "events around every conv launch of the native U-Net executor (separate untimed passes)"}


class OpProbe:
    """The other blocks north_star names, in untimed extra forwards: the fused cross-attention (MFMA-bound), the mask
    head, the geodesic BFS (HBM / latency) and the sampling.  Durations are those of the operators' main KERNELS: two
    events bound to the launch itself inside the native call (gf_dev_op_kernel_events -> hipExtLaunchKernelGGL: the
    dispatch's begin / end timestamps, the quantity a rocprofv3 kernel trace reports) -- not events recorded around the
    Python call, which would include host work between the records (12.7 ms read for a 2.06 ms BFS).  The
    kernels run where the forward launches them (the BFS on the side stream beside the sampling)."""

    NAMES = {"decoder_cross_attn": 1, "mask_head_packed": 2, "mask_head": 2, "geodesic_bfs": 0, "furthest_point_sampling": 3}

    def __init__(self):
        import ctypes

        from geoformer_amd import _lib, pointops

        self.po, self.saved, self.recs = pointops, {}, []
        self.lib, self.ct = _lib.load(), ctypes
        self.pool = []

    def _ev(self):
        e = self.lib.gf_dev_event_create()
        if not e:
            raise RuntimeError("gf_dev_event_create failed")
        self.pool.append(e)
        return e

    def __enter__(self):
        for name in self.NAMES:
            fn = getattr(self.po, name)
            self.saved[name] = fn
            setattr(self.po, name, self._wrap(name, fn))
        return self

    def __exit__(self, *exc):
        for name, fn in self.saved.items():
            setattr(self.po, name, fn)
        for op in set(self.NAMES.values()):
            self.lib.gf_dev_op_kernel_events(op, None, None)

    def _wrap(self, name, fn):
        op = self.NAMES[name]

        def w(*a, **k):
            e0, e1 = self._ev(), self._ev()
            self.lib.gf_dev_op_kernel_events(op, e0, e1)
            r = fn(*a, **k)
            if self.lib.gf_dev_op_kernel_events_taken(op):
                self.recs.append((name, e0, e1, a, r, k))
            else:  # (an early exit without a launch)
                self.lib.gf_dev_op_kernel_events(op, None, None)
            return r

        return w

    def _us(self, e0, e1):
        us = self.ct.c_float()
        if self.lib.gf_dev_event_elapsed_us(e0, e1, self.ct.byref(us)) != 0:
            raise RuntimeError("gf_dev_event_elapsed_us failed")
        return float(us.value)

    def result(self):
        torch.cuda.synchronize()
        acc = {}
        for name, e0, e1, a, r, kw in self.recs:
            us = self._us(e0, e1)
            if name == "decoder_cross_attn":
                B, nq, nc = a[0].shape
                d = a[7].shape[-1]
                w = {"flop": 3 * 2 * nq * nc * B * d * d, "bytes": 4 * (nq * nc * B + (nq + 2 * nc) * B * d + nq * B * d)}
            elif name in ("mask_head_packed", "mask_head"):
                N, C = a[0].shape
                nq = a[3].shape[0]
                w = {"flop": 2 * nq * N * ((C + 3) * C + C), "bytes": 4 * (2 * nq * N + (C + 3) * N)}
                name = "mask_head"
            elif name == "geodesic_bfs":
                n, K = a[0].shape
                nq = a[3].shape[0]
                reached = int((r >= 0).sum())
                # graph rows (12 bytes x 63 entries per vertex), distance write, 8 bytes per frontier entry
                w = {"flop": 0, "bytes": 12 * (K - 1) * n + 4 * nq * n + 8 * reached, "pairs_reached": reached}
            else:
                known = kw.get("known", a[2] if len(a) > 2 else None)
                w = {"flop": 0, "bytes": 0, "picks": int(r.shape[-1]) - (int(known.shape[-1]) if known is not None else 0)}
            t = acc.setdefault(name, {"us": 0.0, "n": 0})
            t["us"] += us
            t["n"] += 1
            for k2, v in w.items():
                t[k2] = t.get(k2, 0) + v
        for e in self.pool:
            self.lib.gf_dev_event_destroy(e)
        self.pool = []
        clock = ("two events bound to the kernel launch inside the native call (hipExtLaunchKernelGGL: the dispatch's "
                 "begin / end timestamps), on the stream the forward launches it on")
        out = {}
        if "decoder_cross_attn" in acc:
            t = acc["decoder_cross_attn"]
            tf = t["flop"] / (t["us"] * 1e-6) / 1e12
            out["roofline_decoder"] = {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None, "kernel": "k_decoder_cross_attn",
                                       # the 16-wave shape runs every product as 6 bf16 MFMAs over the exact three-piece split
                                       # (fp32-accurate, tests/test_gpu_heads.py): against the pipe it executes on
                                       "executed_bf16_tflops": round(6 * tf, 1), "bf16_peak_tflops": 2500.0,
                                       "frac_of_bf16_pipe": round(6 * tf / 2500.0, 4),
                                       "frac_note": "frac = fp32-equivalent flops / fp32 matrix peak (the arithmetic the path "
                                                    "computes); frac_of_bf16_pipe = executed bf16 flops (6 per product) / dense "
                                                    "bf16 peak: the unit it runs on is 37-40 % busy, not 99 %",
                                       "launches": t["n"], "us_per_launch": round(t["us"] / t["n"], 2),
                                       "flop_per_launch": t["flop"] // t["n"],
                                       "formula": "3 * 2 * nq * nc * B * d^2 (SURVEY 8d: the pair MLP's two layers + the value projection)",
                                       "clock": clock}
        if "mask_head" in acc:
            t = acc["mask_head"]
            tf = t["flop"] / (t["us"] * 1e-6) / 1e12
            gb = t["bytes"] / (t["us"] * 1e-6) / 1e9
            out["roofline_mask_head"] = {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None, "kernel": "k_mask_head",
                                         "launches": t["n"], "us_per_launch": round(t["us"] / t["n"], 2),
                                         "flop_per_launch": t["flop"] // t["n"], "bytes_per_launch": t["bytes"] // t["n"],
                                         "hbm_GB/s": round(gb, 1), "hbm_frac": round(gb / HBM_PEAK_GBS, 4),
                                         "formula": "flop 2 * nq * N_fg * (19 * 16 + 16), bytes 4 * (2 * nq * N_fg + 19 * N_fg): "
                                                    "80 flop/B, above the fp32 ridge (20)", "clock": clock}
        if "geodesic_bfs" in acc:
            t = acc["geodesic_bfs"]
            gb = t["bytes"] / (t["us"] * 1e-6) / 1e9
            out["roofline_bfs"] = {"bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(gb / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "k_geodesic_bfs_lds",
                                   "launches": t["n"], "us_per_launch": round(t["us"] / t["n"], 1),
                                   "bytes_per_launch": t["bytes"] // t["n"],
                                   "frontier_entries_per_launch": t["pairs_reached"] // t["n"],
                                   "formula": "12 * 63 * N_fg (graph) + 4 * nq * N_fg (distances) + 8 * sum of frontier sizes",
                                   "note": "a chain of up to 256 dependent hops per query: latency-bound, runs beside the sampling",
                                   "clock": clock}
        if "furthest_point_sampling" in acc:
            t = acc["furthest_point_sampling"]
            out["sampling"] = {"kernel": "k_fps", "launches": t["n"], "us_total_per_forward": None, "picks": t["picks"],
                               "us_per_pick": round(t["us"] / max(t["picks"], 1), 3),
                               "note": "2047 dependent arg-max rounds over <= 50 000 points: serial latency, no roofline",
                               "clock": clock}
        return out


def op_rooflines(model, batches, reps=2):
    with OpProbe() as pr:
        for i, b in enumerate(batches[:reps]):
            np.random.seed(2000 + i)
            with torch.no_grad():
                model(b, 300, training=False)
        return pr.result()


def _timed(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def secondary_train_step_b4(dev, steps=5):
    """BASELINE config 3: one training step of a batch of 4 scenes (150k / 120k / 180k / 100k points), train yaml with
    batch_size 4: forward + criterion + backward + fused Adam, for both epoch regimes (tools/train_dp.py's loop on one
    rank: flat gradient buffer, no exchange)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import train_dp

    from geoformer_amd import scene

    mk = lambda seeds: to_device(scene.make_batch([scene.make_scene(int(n), sd) for n, sd in seeds]), dev)  # noqa: E731
    batches = [mk(((150_000, 50), (120_000, 51), (180_000, 52), (100_000, 53))),
               mk(((140_000, 54), (160_000, 55), (110_000, 56), (130_000, 57)))]
    out = {}
    for tag, epoch in (("full_step", 200), ("prepare_epochs_step", 1)):
        args = train_dp.default_args(steps=steps, warmup=2, batch_size=4, epoch=epoch, prepare_epochs=120, fg_frac=0.4)
        r = train_dp.run(args, dev, batches=batches)
        out[tag] = {"ms_per_step": r["ms_per_step"], "scenes_per_s": r["value"], "steps": steps, "epoch": epoch,
                    "prepare_epochs": 120, "points_per_batch": r["points_per_batch"], "last_loss": r["last_loss"],
                    "n_fg_per_step": r["n_fg_per_step"], "fg_frac_target": 0.4}
        torch.cuda.empty_cache()
    try:
        out.update(backward_rooflines(dev))
    except Exception as e:  # noqa: BLE001  (a measurement leg must not take the line down)
        out["backward_rooflines_error"] = repr(e)
    out["config"] = ("config/geoformer_scannet.yaml with batch_size 4, dec_dropout 0; forward + InstSetCriterion + backward + "
                     "fused Adam; full_step = epoch > prepare_epochs (all heads), prepare_epochs_step = backbone + semantic; "
                     "random-init weights with the background logits shifted so that ~40 % of the points are predicted "
                     "foreground like a real scene (as the eval model; n_fg_per_step is what the steps actually saw)")
    return out


def backward_rooflines(dev, reps=5):
    """north_star's other half (train.py:63-75): the backward kernels of the training step at the batch-4 step's sizes,
    each timed by events BOUND to its launch (gf_dev_op_kernel_events / gf_dev_conv_kernel_events), untimed extra launches:
    the level-1 weight gradient and input gradient of the 16 -> 16 submanifold convolutions over the batch's ~520k voxels
    (HBM-bound by the algorithmic count of SURVEY 8d: 4 flop/B), the cross-attention backward (B = 4 scenes, nq = 128,
    nc = 2048) and the mask-head backward (nq = 128 queries over the 30 000 sampled points of a scene), both priced
    against the fp32 matrix peak with 3x the forward's algorithmic flops (recompute + input gradients + weight
    gradients)."""
    import ctypes

    from geoformer_amd import _lib, pointops, scene, sparse

    lib = _lib.load()

    def bound_us(op, fn, conv_events=False):
        us = []
        for i in range(reps + 2):
            e0, e1 = lib.gf_dev_event_create(), lib.gf_dev_event_create()
            if conv_events:
                lib.gf_dev_conv_kernel_events(e0, e1)
            else:
                lib.gf_dev_op_kernel_events(op, e0, e1)
            fn()
            took = lib.gf_dev_conv_kernel_events_taken() if conv_events else lib.gf_dev_op_kernel_events_taken(op)
            if took and i >= 2:
                v = ctypes.c_float()
                if lib.gf_dev_event_elapsed_us(e0, e1, ctypes.byref(v)) == 0:
                    us.append(float(v.value))
            torch.cuda.synchronize()
            if conv_events:
                lib.gf_dev_conv_kernel_events(None, None)
            else:
                lib.gf_dev_op_kernel_events(op, None, None)
            lib.gf_dev_event_destroy(e0), lib.gf_dev_event_destroy(e1)
        return float(np.mean(us)) if us else None

    out = {}
    clock = "two events bound to the kernel launch (hipExtLaunchKernelGGL), mean of %d launches after 2 warm-up ones" % reps
    b = scene.make_batch([scene.make_scene(int(n), 50 + i) for i, n in enumerate((150_000, 120_000, 180_000, 100_000))])
    c = b["voxel_locs"].to(dev).int().contiguous()
    shape = tuple(int(x) for x in b["spatial_shape"])
    rules = sparse.subm_rules(c, sparse.build_index(c, 4, shape))
    M = int(c.shape[0])
    R = int((rules.nbr[:, :M] >= 0).sum().item())
    x = torch.randn(M, 16, device=dev)
    g = torch.randn(M, 16, device=dev)
    W = torch.randn(27, 16, 16, device=dev) * 0.05
    byt = _conv_bytes(R, M, 27, 16, 16)
    t = bound_us(7, lambda: sparse.conv_wgrad(x, g, rules.nbr, 27, M, rules.ld, gmask=rules.gmask))
    if t:
        out["roofline_wgrad"] = {"bound": "hbm", "achieved": round(byt / (t * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(byt / (t * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                                 "kernel": "k_conv_wgrad_t, level-1 3x3x3 16 -> 16", "us_per_launch": round(t, 2),
                                 "algorithmic_bytes": int(byt), "gflop": round(2 * R * 256 / 1e9, 3), "voxels": M, "rules": R,
                                 "formula": "4 * (R*Cin + M*Cout + K*Cin*Cout) + 8*R: every rule gathers one input row, every "
                                            "gradient row read once, the K*Cin*Cout sums written once", "clock": clock}
    bwd = ("subm", (rules.nbr, rules.gmask, 27, M, rules.ld, rules.steps))
    t = bound_us(None, lambda: sparse.conv_dgrad(g, W, bwd, M), conv_events=True)
    if t:
        out["roofline_dgrad"] = {"bound": "hbm", "achieved": round(byt / (t * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(byt / (t * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                                 "kernel": "k_conv_g16p over the transposed relation (weights W[26-k]^T), level-1 16 -> 16",
                                 "us_per_launch": round(t, 2), "algorithmic_bytes": int(byt), "voxels": M, "rules": R, "clock": clock}
    del x, g, rules
    # cross-attention backward: B = 4, nq = 128, nc = 2048, d = 64
    B, nq, nc, d = 4, 128, 2048, 64
    rn = lambda *sh: torch.randn(*sh, device=dev)  # noqa: E731
    geo_ctx = torch.rand(B, nq, nc, device=dev) * 3
    max_geo = geo_ctx.amax(2).contiguous()
    qloc, cloc = rn(B, nq, 3), rn(B, nc, 3)
    lo, hi = -torch.ones(B, 3, device=dev) * 4, torch.ones(B, 3, device=dev) * 4
    gaussB = rn(3, 32)
    Q1, K1, Kv = (rn(B, nq, d).requires_grad_(), rn(B, nc, d).requires_grad_(), rn(B, nc, d).requires_grad_())
    W1, W2, Wv = (rn(d, d) * 0.1).requires_grad_(), (rn(d, d) * 0.1).requires_grad_(), (rn(d, d) * 0.1).requires_grad_()
    go = rn(B, nq, d)

    def ca_step():
        o = pointops.decoder_cross_attn_train(geo_ctx, max_geo, qloc, cloc, lo, hi, gaussB, Q1, K1, Kv, W1, W2, Wv)
        o.backward(go)

    t = bound_us(4, ca_step)
    if t:
        fl = 9 * 2 * nq * nc * B * d * d
        out["roofline_decoder_bwd"] = {"bound": "mfma", "achieved": round(fl / (t * 1e-6) / 1e12, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": round(fl / (t * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                       "traffic": None, "kernel": "k_decoder_cross_attn_bwd", "us_per_launch": round(t, 2),
                                       "flop_per_launch": fl, "formula": "3 x the forward's 3 * 2 * nq * nc * B * d^2 (recompute, "
                                       "input gradients, weight gradients); B = 4, nq = 128, nc = 2048", "clock": clock}
    # mask-head backward: nq = 128 queries over 30 000 sampled points
    N, C = 30_000, 16
    feat = rn(N, C).requires_grad_()
    params = (rn(nq, C * (C + 3) + 2 * C + 1) * 0.2).requires_grad_()
    coords, qx = rn(N, 3), rn(nq, 3)
    geo = torch.rand(nq, N, device=dev) * 3
    geo[:, ::7] = -1.0
    mx = torch.sqrt(geo.amax(1)).contiguous()
    gl = rn(nq, N)

    def mh_step():
        o = pointops.mask_head_train(feat, params, coords, geo, qx, mx)
        o.backward(gl)

    fwd_fl = 2 * nq * N * ((C + 3) * C + C)
    tf_ = bound_us(5, mh_step)
    tp_ = bound_us(6, mh_step)
    if tf_ and tp_:
        t = tf_ + tp_
        out["roofline_mask_head_bwd"] = {"bound": "mfma", "achieved": round(3 * fwd_fl / (t * 1e-6) / 1e12, 2),
                                         "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "frac": round(3 * fwd_fl / (t * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                                         "kernel": "k_mask_head_bwd_feat + k_mask_head_bwd_param",
                                         "us_per_launch": round(t, 2), "us_feat_kernel": round(tf_, 2), "us_param_kernel": round(tp_, 2),
                                         "flop_per_launch": 3 * fwd_fl, "formula": "3 x the forward's 2 * nq * N * (19 * 16 + 16) "
                                         "(recompute, feature gradients, parameter gradients); nq = 128, N = 30 000", "clock": clock}
    torch.cuda.empty_cache()
    return out


def secondary_few_shot(dev, steps=5, train_leg=True):
    """BASELINE config 4: a 1-way k-shot episode = k full support scenes through process_support, their mean embedding,
    one GeoFormerFS.forward on the S150k query scene (test yaml); k = 1 (the shipped yaml) and 5 (BASELINE.json)."""
    from geoformer_amd import scene
    from geoformer_amd.model import GeoFormerFS, load_config
    from tests.util import synthetic_state_dict

    cfg = load_config("test_geoformer_fs_scannet.yaml")
    m = GeoFormerFS(cfg)
    m.load_state_dict(synthetic_state_dict(m.state_dict(), 2))
    with torch.no_grad():
        m.semantic_linear.bias[3] += 1.0
    m.to(dev)
    m.eval()

    def fsd(sc):
        d = scene.make_batch([sc])
        d["batch_offsets"] = d["offsets"]
        d["support_masks"] = (d["instance_labels"] >= 0).long()
        return to_device(d, dev)

    q = fsd(scene.make_scene(150_000, 1234))
    sups = [fsd(scene.make_scene(130_000, 70 + i)) for i in range(5)]
    out = {}
    for k in (1, 5):
        def episode():
            with torch.no_grad():
                emb = torch.stack([m.process_support(sups[i], training=False) for i in range(k)]).mean(0)
                return m(None, q, training=False, remember=False, support_embeddings=emb)

        dt = _timed(episode, steps)
        out[f"fs_{k}shot"] = {"ms_per_episode": round(dt * 1e3, 2), "episodes_per_s": round(1.0 / dt, 2), "steps": steps,
                              "config": f"config/test_geoformer_fs_scannet.yaml, 1-way {k}-shot: {k} full support scenes "
                                        "(130k points) + one S150k query scene"}
    if train_leg:
        out.update(secondary_few_shot_train(dev))
    with torch.no_grad():
        emb0 = m.process_support(sups[0], training=False)
        embs = torch.cat([emb0 * (0.5 + 0.1 * i) for i in range(10)])
        m(None, q, training=False, remember=False, support_embeddings=emb0)
    dt = _timed(lambda: m.requery_many(q, embs), steps)
    out["fs_requery_x10"] = {"ms_per_requery": round(dt * 1e2, 3), "steps": steps,
                             "config": "10 cached re-queries of the query scene queued together (requery_many, row f4)"}
    return out


def secondary_few_shot_train(dev, steps=4):
    """The training-mode few-shot episode of config/geoformer_fs_scannet.yaml: a batch of 4 query
    scenes, one FULL support scene per query, frozen backbone (42 706 trainable parameters), forward +
    FSInstSetCriterion + backward + Adam."""
    from geoformer_amd import scene
    from geoformer_amd.model import GeoFormerFS, load_config
    from geoformer_amd.model.criterion_fs import FSInstSetCriterion
    from tests.util import synthetic_state_dict

    cfg = load_config("geoformer_fs_scannet.yaml", batch_size=4, dec_dropout=0.0)
    m = GeoFormerFS(cfg)
    m.load_state_dict(synthetic_state_dict(m.state_dict(), 4))
    with torch.no_grad():
        m.semantic_linear.bias[4:] += 1.0  # train fold == cv fold: foreground = classes >= 4
    m.to(dev)
    m.train()
    crit = FSInstSetCriterion(cfg)
    params = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3, fused=True)
    q = scene.make_batch([scene.make_scene(n, 80 + i) for i, n in enumerate((150_000, 120_000, 140_000, 110_000))])
    sup = scene.make_batch([scene.make_scene(n, 90 + i) for i, n in enumerate((130_000, 100_000, 120_000, 110_000))])
    for d in (q, sup):
        d["batch_offsets"] = d["offsets"]
    sup["support_masks"] = (sup["instance_labels"] >= 0).long()
    q, sup = to_device(q, dev), to_device(sup, dev)

    def step():
        np.random.seed(5)
        o = m(sup, q, training=True)
        loss, _ = crit(o, q, 5)
        opt.zero_grad()
        loss.backward()
        opt.step()

    dt = _timed(step, steps)
    return {"fs_train_episode_b4": {"ms_per_step": round(dt * 1e3, 2), "steps": steps,
                                    "trainable_parameters": int(sum(p.numel() for p in params)),
                                    "points": [int(q["locs"].shape[0]), int(sup["locs"].shape[0])],
                                    "config": "config/geoformer_fs_scannet.yaml with batch_size 4: 4 query scenes + one full support "
                                              "scene each, backbone frozen (no_grad), forward + FSInstSetCriterion + backward + Adam"}}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def secondary_test_py_loops(model, batches, dev, args):
    """Two lines in the shape of the reference's own per-scene loop (/root/reference/test.py:52-96), on one GPU:

    synchronous_no_deferral: the headline workload (resident scenes) with NOTHING deferred -- the forward returns the
        proposals and the loop reads the picked masks back to the host before it issues the next scene;
    test_py_shape: per step a pinned HOST batch of a never-before-seen size -> DeviceFeeder (H2D on a copy stream +
        voxelisation on the GPU, one batch ahead like a DataLoader worker) -> forward -> proposals read at once ->
        matrix NMS (util/utils_3d.py:95-141 via geoformer_amd.postprocess) -> D2H of the picked masks, scores, classes.
    """
    from geoformer_amd import feeder, postprocess, scene

    def consume(out, topk=None, final_thresh=0.5):
        """test.py:58-96: proposals -> matrix NMS -> picked masks / scores / classes on the host."""
        ps = out.get("proposal_scores") if isinstance(out, dict) else None
        if ps is None:
            return 0
        if hasattr(ps, "get"):
            ps = ps.get()
        cls_final, scores_final, masks_final = ps
        if isinstance(cls_final, list) or cls_final.shape[0] == 0:
            return 0
        if topk is not None and scores_final.shape[0] > topk:
            keep = torch.topk(scores_final, topk).indices
            cls_final, scores_final, masks_final = cls_final[keep], scores_final[keep], masks_final[keep]
        pick = postprocess.matrix_non_max_suppression(masks_final, scores_final, cls_final, final_score_thresh=final_thresh)
        clusters = masks_final[pick].cpu().numpy()
        scores_final[pick].cpu().numpy()
        cls_final[pick].cpu().numpy()
        return int(clusters.shape[0])

    res = {}
    k = min(args.steps, 16)
    ns = len(batches)
    # -- the headline's resident scenes, strictly one after the other
    picked = 0
    for i in range(ns):
        np.random.seed(1000 + i)
        with torch.no_grad():
            consume(model(batches[i % ns], 300, training=False))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(k):
        np.random.seed(3000 + i)
        with torch.no_grad():
            picked += consume(model(batches[i % ns], 300, training=False))
    torch.cuda.synchronize()
    e1 = time.perf_counter() - t1
    res["synchronous_no_deferral"] = {
        "value": round(k / e1, 3), "unit": "scenes/s", "ms_per_step": round(e1 / k * 1e3, 3), "steps": k,
        "instances_picked_per_scene": round(picked / k, 1),
        "config": "the headline workload, nothing deferred: forward -> proposals read -> matrix NMS -> picked masks, scores "
                  "and classes copied to the host, then the next scene (test.py:56-96 without its file output)"}
    # -- fresh host batches through the feeder
    nfresh = 16
    rs = np.random.RandomState(7)
    NWARM = 4  # untimed scenes per leg: pinned buffers, allocator blocks, and -- with instances -- whatever the first real
    # proposals pay once (the first timed step of round 5's with-instances leg took 38 ms; two warm scenes did not cover it)
    sizes = rs.permutation(np.linspace(0.8, 1.2, nfresh + NWARM) * args.points).astype(int)
    # (the largest scene among the untimed ones: the allocator's biggest blocks exist before the clock starts -- one
    #  39 ms step in sixteen was a fresh device allocation for the first scene larger than all before it)
    j_max = int(np.argmax(sizes))
    sizes[0], sizes[j_max] = sizes[j_max], sizes[0]
    raws = [scene.collate_raw([scene.make_scene(int(n), 7000 + j)]) for j, n in enumerate(sizes)]
    cfg_thresh = model.cfg.TEST_SCORE_THRESH
    for name, forced in (("test_py_shape", False), ("test_py_shape_with_instances", True)):
        # a random-init network scores every proposal ~0.03, so at the yaml's threshold (0.5) nothing reaches the NMS;
        # the second leg lets the 40 best-scored proposals of a scene through (threshold 0, top 40, NMS cut 0) so that
        # the NMS and the copy of the picked masks move a trained net's kind of volume
        model.cfg.TEST_SCORE_THRESH = 0.0 if forced else cfg_thresh
        if forced:
            # (one-time costs of the NMS path -- the intersection kernel's scratch, the framework's sort / matmul set-up --
            #  stay out: the two untimed scenes below may pass no proposal at all)
            wm = (torch.rand((40, int(1.3 * args.points)), device=dev) < 0.01).int()
            postprocess.matrix_non_max_suppression(wm, torch.rand(40, device=dev), torch.randint(4, 13, (40,), device=dev),
                                                   final_score_thresh=0.0)
            wm[:8].cpu()
            del wm
            # (... and whatever the first scene that really passes proposals pays once -- 35 ms in the middle of the timed
            #  steps otherwise, most fresh scenes pass none: the resident scenes go through the same consumer untimed)
            for i in range(ns):
                np.random.seed(2000 + i)
                with torch.no_grad():
                    consume(model(batches[i], 300, training=False), topk=40, final_thresh=0.0)
            torch.cuda.synchronize()
        picked, nsteps, t1 = 0, 0, None
        per_step = []
        try:
            for j, batch in enumerate(feeder.DeviceFeeder(raws, dev, reserve_points=int(1.3 * args.points))):
                if j == NWARM:  # (untimed scenes: the feeder's pinned buffers and the allocator's blocks exist)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                ts = time.perf_counter()
                np.random.seed(4000 + j)
                with torch.no_grad():
                    out = model(batch, 300, training=False)
                    n_inst = consume(out, topk=40, final_thresh=0.0) if forced else consume(out)
                if j >= NWARM:
                    picked += n_inst
                    nsteps += 1
                    per_step.append(round((time.perf_counter() - ts) * 1e3, 2))
            torch.cuda.synchronize()
            e1 = time.perf_counter() - t1
        finally:
            model.cfg.TEST_SCORE_THRESH = cfg_thresh
        res[name] = {
            "value": round(nsteps / e1, 3), "unit": "scenes/s", "ms_per_step": round(e1 / nsteps * 1e3, 3), "steps": nsteps,
            "ms_per_step_median": round(float(np.median(per_step)), 3) if per_step else None,
            "points": [int(r["locs"].shape[0]) for r in raws[NWARM:]],
            "instances_picked_per_scene": round(picked / max(nsteps, 1), 1),
            "ms_forward_and_consume_per_step": per_step,
            "config": "per step: pinned host batch of a never-before-seen size -> geoformer_amd.feeder.DeviceFeeder (H2D on a "
                      "copy stream, voxelisation on the GPU, one batch ahead; its ~0.8 ms of staging runs on the consumer's "
                      "thread) -> forward -> proposals read at once -> matrix NMS -> D2H of the picked masks; PCIe-inclusive, "
                      "nothing deferred (the shape of test.py:52-96)" +
                      ("; score threshold 0, the 40 best proposals kept, NMS cut 0: a random-init net passes nothing at the "
                       "yaml's 0.5" if forced else "; yaml thresholds: a random-init net passes no proposal, the NMS and the "
                       "mask copy move nothing")}
    return res


def self_launch(args, argv):
    """`bench.py --gpus N` outside a torchrun environment: start the N ranks as a CHILD process, relay what rank 0
    prints, exit with the child's code.  This process makes no HIP call at all (not even a device count, which would
    initialise the runtime here): a rank without a device fails with its own message."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            print(line, flush=True)
    return r.returncode


def plumbing_test(args):
    """No GPU work: the launch / rendezvous / timing / one-line-from-rank-0 plumbing over gloo (tests/test_parallel_gloo.py)."""
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "plumbing test (no GPU work)", "value": round(world * args.steps / elapsed, 3),
                          "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "plumbing_test": True}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


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


def cpu_forward(batch, bias_shift, seed, threads, timed=3):
    """Eval forwards of `batch` through the build's model on the host with the oracle's C operators: one untimed (page
    faults, OpenMP team start-up, the oracle's lazily built tables), then `timed` timed ones.  Returns the last output, the
    list of wall times, the stage seconds of the MEDIAN run and the OpenMP thread count actually set."""
    from oracle import cpu_backend
    from oracle import oracle as orc

    torch.set_num_threads(threads)
    L = orc.lib()
    L.orc_set_threads.restype = int
    omp = int(L.orc_set_threads(threads))
    runs = []
    with cpu_backend.installed(), torch.no_grad():
        m = build_model("cpu", bias_shift=bias_shift)
        np.random.seed(seed)
        m(batch, 300, training=False)  # untimed
        for _ in range(max(1, timed)):
            st = StageTimer(m)
            try:
                np.random.seed(seed)
                t = time.perf_counter()
                out = m(batch, 300, training=False)
                dt = time.perf_counter() - t
            finally:
                st.close()
            runs.append((dt, st.stages()))
    order = sorted(range(len(runs)), key=lambda i: runs[i][0])
    med = runs[order[len(order) // 2]]
    return out, [r[0] for r in runs], med[1], omp


def cpu_baseline_and_parity(model, batch, dev, seed=4321):
    """cpu_baseline (the host forward of scene 0, all host cores) and parity_s150k (its outputs against the GPU
    forward of the same scene under the same numpy seed)."""
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, int(os.environ.get("GF_CPU_BASELINE_THREADS", "64"))))
    host_batch = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in batch.items()}
    outc, dts, stages, omp = cpu_forward(host_batch, model._bench_bias_shift, seed, threads)
    dt = float(np.median(dts))
    n = int(host_batch["locs"].shape[0])
    base = {"value": round(1.0 / dt, 5), "unit": "scenes/s", "cores": omp, "kind": "port",
            "sample": f"eval forward of benchmark scene 0 ({n} points, N_fg={int(outc['fg_idxs'].shape[0])}): one untimed, "
                      f"then {len(dts)} timed, MEDIAN quoted; oracle C operators on {omp} OpenMP threads, torch modules on "
                      f"{torch.get_num_threads()} threads ({cores} host cores; GF_CPU_BASELINE_THREADS sets both)",
            "seconds": round(dt, 2), "seconds_per_run": [round(x, 2) for x in dts], "stage_seconds": stages}
    np.random.seed(seed)
    with torch.no_grad():
        outg = model(batch, 300, training=False)
    torch.cuda.synchronize()
    par = compare_outputs(outg, outc)
    # the same comparison under weights with trained-net-like activation scales: 1e-4 ABSOLUTE, nothing relative
    cal = None
    if os.environ.get("GF_BENCH_CALIBRATED", "1") != "0":
        from oracle import cpu_backend
        from tests.util import calibrated_benchmark_state

        state, raw_scale = calibrated_benchmark_state(host_batch)
        with cpu_backend.installed(), torch.no_grad():
            mc = build_model("cpu")
            mc.load_state_dict(state)
            np.random.seed(seed)
            oc = mc(host_batch, 300, training=False)
        mg = build_model(dev)
        mg.load_state_dict(state)
        np.random.seed(seed)
        with torch.no_grad():
            og = mg(batch, 300, training=False)
        torch.cuda.synchronize()
        cal = compare_outputs(og, oc, ulps=0)
        cal["weights"] = ("BatchNorm statistics = this scene's activation statistics, last semantic layer scaled to class scores "
                          "of ~ +-8, controller scaled to mask logits of ~ +-4 (tests/util.calibrated_benchmark_state; the "
                          "random controller gave %.0f)" % raw_scale)
        cal["n_fg"] = int(og["fg_idxs"].shape[0])
        del mg
    return base, par, cal


def compare_outputs(outg, outc, ulps=PARITY_ULPS):
    """max-abs differences of a GPU forward against the host forward of the same scene, with BOTH bounds evaluated:
    north_star's 1e-4 absolute (`within_1e-4_abs`) and max(1e-4, ulps * 2^-23 * the tensor's largest magnitude)
    (`within_tolerance`; ulps = 0: the absolute bound alone)."""
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
    eps = float(np.finfo(np.float32).eps)
    par["tolerance"] = ("1e-4 absolute" if ulps == 0 else
                        f"max(1e-4, {ulps} * 2^-23 * the tensor's largest magnitude (*_scale)): the difference of two fp32 "
                        "evaluations, each held to 32 epsilons of the float64 arbiter in tests/test_gpu_fullsize.py (DESIGN.md 2)")
    bad, bad_abs = [], []
    for key, scale_key in (("semantic_scores_maxabs", "semantic_scores_scale"), ("mask_logits_maxabs", "mask_logits_scale"),
                           ("cls_logits_maxabs", None), ("proposal_scores_maxabs", None)):
        if key in par:
            bound = max(1e-4, ulps * eps * (par[scale_key] if scale_key else 1.0))
            par[key.replace("_maxabs", "_bound")] = bound
            if not par[key] <= bound:
                bad.append(key)
            if not par[key] <= 1e-4:
                bad_abs.append(key)
    if par["fg_idxs_xor"] == 0 and par.get("proposals", [0, 0])[0] != par.get("proposals", [0, 0])[1]:
        bad.append("proposals")
        bad_abs.append("proposals")
    par["within_tolerance"] = not bad
    par["violations"] = bad
    par["within_1e-4_abs"] = not bad_abs and par["fg_idxs_xor"] == 0
    par["violations_1e-4_abs"] = bad_abs
    return par


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--points", type=int, default=150_000)
    ap.add_argument("--scenes", type=int, default=8, help="resident scenes the steps rotate over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-ahead", action="store_true",
                    help="rulebooks behind the caller's stream like every other launch (default: ahead of it, see --help of DESIGN.md 7)")
    ap.add_argument("--staggered", dest="pipeline", action="store_true",
                    help="two staggered scenes in flight (geoformer_amd/serving.py) instead of one scene at a time")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false", help="(default) one scene at a time")
    ap.set_defaults(pipeline=False)
    ap.add_argument("--plumbing-test", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))
    if args.plumbing_test:
        sys.exit(plumbing_test(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} inside a torchrun environment of WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP operators have no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        # N ranks share one host: every rank's framework threads get their share of the cores (a forward's host side is
        # ~2.6 ms of Python per scene; 8 ranks x the default thread pool would fight for the same cores)
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // world))

    from geoformer_amd import scene, serving

    if os.environ.get("GF_CONV_CHUNKS"):  # dev knob: waves of the level-1 conv kernel (include/geoformer_hip_dev.h)
        from geoformer_amd import sparse

        sparse.dev_conv_chunks(int(os.environ["GF_CONV_CHUNKS"]))
    # every rank gets its own scenes: replicas of the same workload
    ns = max(1, args.scenes)
    batches = [to_device(scene.make_batch([scene.make_scene(args.points, 1234 + rank * ns + i)]), dev)
               for i in range(ns)]
    if not args.no_ahead:
        # the scenes are resident before the clock starts (the bench contract): nothing produces their tensors any more,
        # which a batch says with an empty event list -- the model then builds a scene's rulebooks on the executor's side
        # stream without waiting for the previous scene's tail on the caller's (GeoFormer._inputs_ahead, gf_unet_fwd_ahead)
        torch.cuda.synchronize()
        for b in batches:
            b["inputs_event"] = ()
    model = build_model(dev, probe_batch=batches[0])
    Ms = [int(b["voxel_locs"].shape[0]) for b in batches]
    probe = ConvProbe(batches)

    class Loop:
        """One scene per step.  Default: one scene at a time on the current stream, scene i's proposals collected after
        scene i+1 is issued (what the unchanged test.py:60-110 runs).  pipeline=True: two scenes in flight, staggered
        (geoformer_amd/serving.py).  Every scene's outputs, proposals included, are complete when `finish` returns."""

        def __init__(self, pipeline=None, scenes=None):
            self.prev = None
            self.last = None
            self.stag = {}
            self.pipeline = args.pipeline if pipeline is None else pipeline
            self.scenes = batches if scenes is None else scenes

        def step(self, i, m=model):
            np.random.seed(1000 + i)
            b = self.scenes[i % len(self.scenes)]
            if not self.pipeline:
                with torch.no_grad():
                    out = m(b, 300, training=False, defer_proposals=True)
                self._collect()
                self.prev = self.last = out
                return out
            st = self.stag.get(id(m))
            if st is None:
                st = self.stag[id(m)] = serving.StaggeredForward(m, dev)
            for out in st.submit(b, seed=1000 + i):
                self.last = out

        def _collect(self):
            if self.prev is not None and not isinstance(self.prev.get("proposal_scores"), (tuple, type(None))):
                self.prev["proposal_scores"] = self.prev["proposal_scores"].get()
            self.prev = None

        def finish(self):
            self._collect()
            for st in self.stag.values():
                for out in st.drain():
                    self.last = out

        def timed(self, first, k, m=model):
            """k steps between two device synchronisations -> seconds."""
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(k):
                self.step(first + i, m)
            self.finish()
            torch.cuda.synchronize()
            return time.perf_counter() - t1

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
    from geoformer_amd import _lib as _gl

    _gl.host_wait_s[0] = 0.0
    _gl.load().gf_dev_host_wait_ns(1)
    t0 = time.perf_counter()
    for i in range(args.steps):
        probe.arm(i % PROBE_EVERY == 0)
        out = step(args.warmup + i)
    loop.finish()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    probe.close()
    # host_busy: wall time of the loop's issue phase minus the time this thread was blocked in the package's own waits
    # (foreground count, proposals, the executor's two voxel-count read-backs): what the host itself costs per scene
    host_busy = t_issue - _gl.host_wait_s[0] - _gl.load().gf_dev_host_wait_ns(0) * 1e-9
    busy_all = [host_busy]
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        hb = torch.zeros(world, device=dev, dtype=torch.float64)
        hb[rank] = host_busy
        dist.all_reduce(hb)
        busy_all = [float(x) for x in hb.tolist()]

    dp = None
    if world > 1 and not args.no_secondary:
        # BASELINE config 5: every rank a batch of 4 scenes, bucketed gradient all-reduce over RCCL (all ranks take part)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import train_dp

        dp = train_dp.run(train_dp.default_args(steps=5, warmup=2, batch_size=4, epoch=200, prepare_epochs=120, fg_frac=0.4), dev)
        torch.cuda.empty_cache()
    if rank == 0:
        n_fg = int(loop.last["fg_idxs"].shape[0])
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
                                   "random-init weights; fp32 throughout, the decoder cross-attention and the mask head "
                                   "on the bf16 matrix pipe over the EXACT three-piece bf16 split of their fp32 operands "
                                   "(fp32-accurate against float64: tests/test_gpu_heads.py); " +
                                   ("two scenes in flight on two streams, staggered (geoformer_amd/serving.py): scene i's "
                                    "decoder + mask head run under scene i+1's sampling / BFS stretch, scene i+1's backbone "
                                    "starts when scene i's stretch has ended; all scenes complete inside the timed region"
                                    if args.pipeline else
                                    "one scene at a time as in the reference's test.py:60-110, proposals of scene i "
                                    "collected after scene i+1 is issued; all scenes complete inside the timed region" +
                                    ("" if args.no_ahead else
                                     "; the scenes are resident and say so (inputs_event = ()): a forward's voxel features and "
                                     "rulebooks -- functions of its input alone -- are queued on the U-Net executor's side "
                                     "stream, not behind the previous scene's tail (gf_unet_fwd_ahead; --no-ahead: ~1.4 % less)")),
                       "points": [int(b["locs"].shape[0]) for b in batches], "voxels": Ms, "n_fg_last": n_fg,
                       "parallelism": f"replicas x{world}"},
            "roofline": None,
            "host_busy_ms_per_step": {"per_rank": [round(x / args.steps * 1e3, 3) for x in busy_all],
                                      "threads_per_rank": torch.get_num_threads(),
                                      "note": "wall time of the timed loop's issue phase minus the time the host thread was blocked "
                                              "in waits (foreground count, proposals, the U-Net executor's voxel counts): the "
                                              "Python + launch cost of a scene; N ranks share one host's cores"},
        }
        l1 = probe.result()
        fam = res["roofline_convs"] = all_convs_roofline(model, batches)
        if l1 is not None and l1.get("traffic"):
            # what the level-1 launch really moves at the memory side (PMC: FETCH_SIZE x 2 + WRITE_SIZE), against the same clock
            l1["actual_hbm_frac"] = round(l1["traffic"] / (l1["us_per_launch"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        if fam is not None:
            # the headline object is the FAMILY -- the 71 sparse convolutions of a forward, the north star's ">= 50 % of the
            # binding roofline" -- with the level-1 16 -> 16 launch (the family's best kernel, 2-3 % of device time) beside it
            res["roofline"] = {
                "bound": "hbm", "achieved": fam["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fam["frac"],
                "traffic": PMC_FAMILY.get("bytes_per_forward"),
                "traffic_note": "HBM-side bytes of the family PER FORWARD, like achieved (2 FETCH_SIZE + WRITE_SIZE in KiB over "
                                "every k_conv_* launch and k_concat2_idn: " + str(PMC_FAMILY.get("source"))[:200] + ")",
                "actual_hbm_frac": (round(PMC_FAMILY["bytes_per_forward"] / (fam["us_per_forward"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                                    if PMC_FAMILY.get("bytes_per_forward") else None),
                **_family_rocprof(fam),
                "kernel": "the sparse-convolution family: all 71 launches of a forward (k_conv_g16p / k_conv_lw / k_conv_os / "
                          "k_conv_flat / k_conv_pair), sum of SURVEY 8(d) algorithmic bytes / sum of launch durations "
                          "(events around every launch, untimed extra passes: roofline_convs)",
                "us_per_forward": fam["us_per_forward"], "algorithmic_bytes": fam["algorithmic_bytes_per_forward"],
                "compulsory_bytes": fam["compulsory_bytes_per_forward"], "compulsory_frac": fam["compulsory_frac"],
                "compulsory_note": "4 (M_in Cin + M_out Cout) per launch: every element read / written once (SURVEY 8d)",
                "level1_16to16": l1,
            }
        else:
            res["roofline"] = l1
        res.update(op_rooflines(model, batches))
        if dp is not None:
            res.setdefault("secondary", {})["train_dp_step"] = dp
        if not args.no_secondary and world == 1:
            m128 = build_model(dev, bias_shift=model._bench_bias_shift, cfg_name="geoformer_scannet.yaml")
            k = min(args.steps, 16)
            for i in range(max(min(args.warmup, 4), ns)):  # every scene once: one-time costs per (model, scene) stay out
                step(i, m128)
            loop.finish()
            e1 = loop.timed(100, k, m128)
            res["secondary"] = {"nq128_train_yaml_eval_forward": {
                "value": round(k / e1, 3), "unit": "scenes/s", "ms_per_step": round(e1 / k * 1e3, 3), "steps": k,
                "config": "config/geoformer_scannet.yaml (nq=128), same scenes, one GPU, same loop as the headline"}}
            del m128
            torch.cuda.empty_cache()
            # the headline workload through the other loop (every scene warmed once in it first)
            other = Loop(pipeline=not args.pipeline)
            for i in range(ns):
                other.step(i)
            other.finish()
            e1 = other.timed(200, k)
            res["secondary"]["one_scene_at_a_time" if args.pipeline else "staggered_two_in_flight"] = {
                "value": round(k / e1, 3), "unit": "scenes/s", "ms_per_step": round(e1 / k * 1e3, 3), "steps": k,
                "config": ("the headline workload one scene at a time" if args.pipeline else
                           "the headline workload through geoformer_amd/serving.py: two scenes in flight on two streams, "
                           "staggered (--staggered makes it the headline); every scene seen once before the timed steps")}
            # a serving workload: every timed step a scene of a size the process has never seen (no per-size cache, no
            # allocator block of the right size), through the headline's loop
            nfresh = 24
            rs = np.random.RandomState(99)
            sizes = rs.permutation(np.linspace(0.72, 1.28, nfresh + 2) * args.points).astype(int)
            fresh = [to_device(scene.make_batch([scene.make_scene(int(n), 5000 + j)]), dev) for j, n in enumerate(sizes)]
            # serving warm-up (untimed): the allocator gets blocks for the largest scene the service admits
            model.reserve_for(max(int(1.3 * args.points), 250_000))  # the yaml's max_npoint: what the dataset admits
            fl = Loop(scenes=fresh)
            fl.step(0), fl.step(1)
            fl.finish()
            e1 = fl.timed(2, nfresh)
            res["secondary"]["fresh_scenes"] = {
                "value": round(nfresh / e1, 3), "unit": "scenes/s", "ms_per_step": round(e1 / nfresh * 1e3, 3), "steps": nfresh,
                "points": [int(b["locs"].shape[0]) for b in fresh[2:]],
                "config": f"{nfresh} timed steps, every one a never-before-seen scene ({int(sizes.min())}-{int(sizes.max())} "
                          "points, all sizes different; warm-up: GeoFormer.reserve_for(250 000, the yaml's max_npoint) -- two forwards at the "
                          "service's size bound, so the allocator already holds blocks for the large buffers -- and two other fresh scenes), resident in "
                          "HBM, same model and loop as the headline"}
            del fresh, fl
            torch.cuda.empty_cache()
            res["secondary"].update(secondary_test_py_loops(model, batches, dev, args))
            res["secondary"]["train_step_b4"] = secondary_train_step_b4(dev)
            res["secondary"].update(secondary_few_shot(dev))
            torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"], res["parity_s150k"], cal = cpu_baseline_and_parity(model, batches[0], dev)
            if cal is not None:
                res["parity_s150k_calibrated"] = cal
        print(json.dumps(res), flush=True)
        for pk in ("parity_s150k", "parity_s150k_calibrated"):
            if pk in res and not res[pk]["within_tolerance"]:
                print(f"bench.py: {pk} outside its stated tolerance: " + ", ".join(res[pk]["violations"]),
                      file=sys.stderr, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
