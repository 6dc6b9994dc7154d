"""ctypes binding of libgeoformer_hip.so (the C ABI declared in include/geoformer_hip.h).

There is NO CPU fallback: if the library is missing or a call fails this module raises.
PyTorch is only used by callers for device memory and streams; pointers cross the boundary
as plain integers.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_size_t, c_uint, c_void_p

from ._build import LIB_PATH

LIB_PATH = os.environ.get("GF_LIB_PATH", LIB_PATH)  # dev knob: load an experimental build of the library

_lib = None


class GeoFormerHipError(RuntimeError):
    pass


def _declare(lib):
    P, I, F = c_void_p, c_int, c_float
    sig = {
        "gf_abi_version": (I, []),
        "gf_last_error": (c_char_p, []),
        "gf_index_words": (c_size_t, [I, I, I, I]),
        "gf_index_scratch_bytes": (c_size_t, [c_size_t]),
        "gf_index_build": (I, [P, I, P, I, I, I, I, P, P, P, P, P]),
        "gf_rules_subm3": (I, [P, I, P, I, I, I, P, P, P, P, I, P, P, P]),
        "gf_rules_steps_words": (c_size_t, [I]),
        "gf_rules_down2": (I, [P, I, P, I, I, I, I, P, P, P, P, P, P, I, P, P, P, I, P, P, P]),
        "gf_rules_down2_chain_plan": (I, [I, I, I, I, I, I, P, P, P, P, P]),
        "gf_rules_down2_chain": (I, [P, I, I, I, I, I, I, P, P, P]),
        "gf_conv_packed_floats": (c_size_t, [I, I, I]),
        "gf_conv_pack_weights": (I, [P, I, I, I, P, P]),
        "gf_conv_pack_weights_t": (I, [P, I, I, I, I, P, P]),
        "gf_conv_fwd": (I, [P, P, P, P, P, I, I, I, I, I, I, P, P, P, P, P, P, P]),
        "gf_feeder_create": (P, [I]),
        "gf_feeder_submit": (I, [P, P]),
        "gf_feeder_wait_issued": (I, [P, I]),
        "gf_feeder_wait_copied": (I, [P, I]),
        "gf_feeder_wait_head": (I, [P, I]),
        "gf_feeder_destroy": (I, [P]),
        "gf_conv_dual_supported": (I, [I, I, I, I, I]),
        "gf_rules_flat_words": (c_size_t, [I, I]),
        "gf_rules_flat_steps": (I, [P, P, I, I, I, I, P, P]),
        "gf_conv_fwd_flat": (I, [P, P, P, P, P, P, I, I, I, I, I, I, P, P, P, P, P, P, P, P]),
        "gf_conv_fwd_dual": (I, [P, P, P, P, P, I, I, I, I, I, I, P, P, P, P, P, P, P, P]),
        "gf_dev_conv_fwd_timed": (I, [P, P, P, P, P, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P]),
        "gf_dev_conv_knobs_g16": (I, [I, I, I, I]),
        "gf_dev_conv_chunks": (I, [I]),
        "gf_dev_conv_g16p_wpb": (I, [I]),
        "gf_dev_conv_knobs": (I, [I, I, I, I, I]),
        "gf_dev_conv_knob_flat": (I, [I, I]),
        "gf_dev_conv_knob_lw": (I, [I, I]),
        "gf_resblock_fwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P, P, P, P, P, P, P, P]),
        "gf_conv_wgrad": (I, [P, P, P, I, I, I, I, I, P, P]),
        "gf_conv_wgrad_masked": (I, [P, P, P, P, I, I, I, I, I, P, P]),
        "gf_conv_wgrad_masked_acc": (I, [P, P, P, P, I, I, I, I, I, P, P]),
        "gf_lsap": (I, [P, I, I, P, P, P, P, P, P]),
        "gf_pair_losses_sums_floats": (c_size_t, [I]),
        "gf_pair_losses_fwd": (I, [P, P, P, I, I, I, P, P, P, P]),
        "gf_pair_losses_bwd": (I, [P, P, P, P, I, I, I, P, P, P, P]),
        "gf_unet_ws_bytes": (c_size_t, [P, I, I, I, I, I]),
        "gf_dev_unet_probe": (I, [I]),
        "gf_dev_host_wait_ns": (ctypes.c_ulonglong, [I]),
        "gf_dev_unet_probe_read": (I, [I, P, P]),
        "gf_dev_unet_probe_read2": (I, [I, P, P, P]),
        "gf_dev_conv_kernel_events": (I, [P, P]),
        "gf_dev_conv_kernel_events_taken": (I, []),
        "gf_dev_op_kernel_events": (I, [I, P, P]),
        "gf_dev_op_kernel_events_taken": (I, [I]),
        "gf_dev_event_create": (P, []),
        "gf_dev_event_destroy": (I, [P]),
        "gf_dev_event_elapsed_us": (I, [P, P, P]),
        "gf_unet_fwd": (I, [P, P, P, I, I, I, I, I, P, c_size_t, P, P, P, P]),
        "gf_unet_fwd_phased": (I, [P, P, P, I, I, I, I, I, P, c_size_t, P, P, P, P, P, I, P, P]),
        "gf_unet_fwd_ahead": (I, [P, P, P, I, I, I, I, I, P, c_size_t, P, P, P, P, P, I]),
        "gf_voxelize_fp": (I, [P, P, I, I, I, I, P, P]),
        "gf_voxelize_bp": (I, [P, P, I, I, I, I, P, P]),
        "gf_gather_points": (I, [P, P, I, I, I, I, P, P]),
        "gf_gather_points_grad": (I, [P, P, I, I, I, I, P, P]),
        "gf_group_points": (I, [P, P, I, I, I, I, I, P, P]),
        "gf_group_points_grad": (I, [P, P, I, I, I, I, I, P, P]),
        "gf_ball_query": (I, [P, P, I, I, I, F, I, P, P]),
        "gf_fps_scratch_bytes": (c_size_t, [I]),
        "gf_furthest_point_sampling": (I, [P, I, I, I, P, P, P]),
        "gf_furthest_point_sampling_resume": (I, [P, I, I, I, I, P, P, P]),
        "gf_knn_scratch_bytes": (c_size_t, [I]),
        "gf_knn_radius": (I, [P, I, I, F, I, P, P, P, P, P]),
        "gf_knn_error_flag": (P, [P, I]),
        "gf_geodesic_bfs": (I, [P, P, P, I, I, P, I, F, I, P, P, P, P]),
        "gf_geodesic_bfs_cfg": (I, [P, P, P, I, I, P, I, F, I, P, P, P, c_size_t, I, P]),
        "gf_geodesic_bfs_queue_words": (c_size_t, [I]),
        "gf_dev_cross_attn_bf3": (I, [I]),
        "gf_dev_bfs_qcap_max": (I, [I]),
        "gf_mask_head": (I, [P, P, P, P, P, P, P, P, P, I, I, I, P, P]),
        "gf_mask_head_packed": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, P, P]),
        "gf_mask_head_episodes": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, P, P]),
        "gf_mask_head_split_bytes": (c_size_t, [I]),
        "gf_mask_head_bwd": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, P, P, P, P]),
        "gf_mask_head_bwd_episodes": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, P, P, P]),
        "gf_mask_head_bwd_scratch_floats": (c_size_t, [I, I]),
        "gf_softmax_dim1_fwd": (I, [P, I, I, I, F, P, P]),
        "gf_softmax_dim1_bwd": (I, [P, P, I, I, I, F, P, P]),
        "gf_pointwise_mlp": (I, [P, I, I, P, P, P, P, P, P, P]),
        "gf_pointwise_mlp_rows": (I, [P, P, I, I, P, P, P, P, P, P, P]),
        "gf_group_mlp_max": (I, [P, I, I, I, I, P, P, P, P, P, P, P]),
        "gf_ball_query_centres": (I, [P, P, I, I, I, F, I, P, P, P]),
        "gf_sa_group_mlp_max": (I, [P, P, P, I, I, I, I, F, I, I, I, I, P, P, P, P, P, P, P, P, P, I, P]),
        "gf_ball_query_grid": (I, [P, I, P, P, I, F, I, P, I, P, P, P]),
        "gf_point_grid_build": (I, [P, I, F, P, P]),
        "gf_decoder_token_state_bytes": (c_size_t, [I, I]),
        "gf_decoder_token_stage": (I, [P, P, P, I, I, I, I, I, P, P, P, P, P, P]),
        "gf_mask_intersections_scratch_bytes": (c_size_t, [I, I]),
        "gf_mask_intersections": (I, [P, I, I, P, P, P]),
        "gf_voxelize_idx_scratch_bytes": (c_size_t, [I]),
        "gf_voxelize_idx_count": (I, [P, I, I, I, P, P, P, P]),
        "gf_voxelize_idx_fill": (I, [P, I, I, I, P, P, I, I, P, P, P]),
        "gf_host_legacy_choice": (I, [P, P, c_longlong, c_longlong, P]),
        "gf_host_legacy_prefetch": (I, [P, I, c_longlong]),
        "gf_host_draw_sample": (I, [P, P, c_longlong, c_longlong, P, c_longlong, P, P, P, P, I, P, P, P]),
        "gf_fg_scratch_bytes": (c_size_t, [I]),
        "gf_fg_select": (I, [P, I, I, I, I, P, P, P, P, I, P, P, P, P, P, P, P, P, P]),
        "gf_host_wait_word": (I, [P, I, c_longlong]),
        "gf_proposal_stats": (I, [P, P, P, I, I, I, F, F, I, I, P, P, P, P, P]),
        "gf_proposal_stats_fs": (I, [P, P, I, I, F, F, I, F, P, P, P, P]),
        "gf_proposal_scatter": (I, [P, P, I, I, P, F, I, P, P]),
        "gf_relpos_prepare": (I, [P, P, I, I, I, P, P, P]),
        "gf_proposal_select": (I, [P, P, P, I, P, P, P, P, P]),
        "gf_backbone_transformer_scratch_bytes": (c_size_t, [I]),
        "gf_backbone_transformer_num_params": (I, [I]),
        "gf_backbone_transformer": (I, [P, P, P, I, I, I, I, P, P, P, P]),
        "gf_decoder_pre_train_save_bytes": (c_size_t, [I, I]),
        "gf_decoder_pre_train_work_bytes": (c_size_t, [I, I]),
        "gf_decoder_pre_grad_floats": (c_longlong, []),
        "gf_decoder_pre_train_fwd": (I, [P, P, I, I, P, F, c_uint, I, P, P, P, P]),
        "gf_decoder_pre_train_bwd": (I, [P, P, P, P, P, I, I, P, F, c_uint, I, P, P, P, P, P, P]),
        "gf_decoder_post_train_save_bytes": (c_size_t, [I, I, I]),
        "gf_decoder_post_train_work_bytes": (c_size_t, [I, I, I]),
        "gf_decoder_post_grad_floats": (c_longlong, [I]),
        "gf_decoder_post_train_fwd": (I, [P, P, I, I, I, P, F, c_uint, I, P, P, P, P]),
        "gf_decoder_post_train_bwd": (I, [P, P, P, P, I, I, I, P, F, c_uint, I, P, P, P, P, P, P]),
        "gf_backbone_transformer_train_save_bytes": (c_size_t, [I, I]),
        "gf_backbone_transformer_train_work_bytes": (c_size_t, [I, I, I]),
        "gf_backbone_transformer_grad_floats": (c_longlong, [I, I]),
        "gf_backbone_transformer_train_fwd": (I, [P, P, I, I, I, I, P, F, c_uint, P, P, P]),
        "gf_backbone_transformer_train_bwd": (I, [P, P, I, I, I, I, P, F, c_uint, P, P, P, P, P]),
        "gf_decoder_wpack_floats": (c_size_t, []),
        "gf_decoder_pack_weights": (I, [P, P, P, P, P]),
        "gf_decoder_cross_attn": (I, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P, P, P, P]),
        "gf_decoder_cross_attn_cfg": (I, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P, P, P, I, P]),
        "gf_decoder_cross_attn_bwd_scratch_floats": (c_size_t, [I, I, I]),
        "gf_decoder_cross_attn_bwd": (I, [P] * 16 + [I, I, I, I] + [P] * 6),
        "gf_bn_train_scratch_floats": (c_size_t, [I, I]),
        "gf_bn_relu_train_fwd": (I, [P, I, I, P, P, F, F, I, P, P, P, P, P, P, P]),
        "gf_bn_relu_train_bwd": (I, [P, P, P, I, I, P, P, P, I, P, P, P, P, P]),
        "gf_bn_relu_train_bwd_add": (I, [P, P, P, I, I, P, P, P, I, P, P, P, P, P, P]),
        "gf_unet_train_scratch_floats": (c_size_t, [P, I, P]),
        "gf_unet_train_fwd": (I, [P, I, I, P, P, P, P, P, P]),
        "gf_unet_train_bwd": (I, [P, I, I, P, P, P, P, P, P, P, P]),
        "gf_bn_train_cl_scratch_floats": (c_size_t, [I, I, c_longlong]),
        "gf_bn_relu_train_cl_fwd": (I, [P, I, I, c_longlong, P, P, F, F, I, P, P, P, P, P, P, P]),
        "gf_bn_relu_train_cl_bwd": (I, [P, P, P, I, I, c_longlong, P, P, P, I, P, P, P, P, P]),
        "gf_sec_op": (I, [I, P, P, I, I, P, P]),
        "gf_roipool_fp": (I, [P, P, I, I, P, P, P]),
        "gf_roipool_bp": (I, [P, P, I, I, P, P]),
        "gf_get_iou": (I, [P, P, P, P, I, I, P, P]),
        "gf_ballquery_batch_p_scratch_bytes": (c_size_t, [I]),
        "gf_ballquery_batch_p": (I, [P, P, P, I, I, F, P, P, P, P, P]),
        "gf_bfs_cluster_host": (I, [P, P, P, I, I, P, P, P, P]),
        "gf_three_nn": (I, [P, P, I, I, I, P, P, P]),
        "gf_three_interpolate": (I, [P, P, P, I, I, I, I, P, P]),
        "gf_three_interpolate_grad": (I, [P, P, P, I, I, I, I, P, P]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return sig


EXPORTS = None


def _check_hw_queues(torch):
    """The forward runs its latency-bound kernels beside each other on several streams; with the HIP runtime's default
    of 4 hardware queues streams alias and those kernels serialise (geoformer_amd/__init__.py).  The runtime reads
    GPU_MAX_HW_QUEUES once, at its first call: nothing can be fixed from here, so say so."""
    import warnings

    from . import HW_QUEUES_MIN, HW_QUEUES_WANTED, hw_queues_setting

    have = hw_queues_setting()
    if have is not None and have >= HW_QUEUES_MIN:
        return
    up = torch.cuda.is_initialized()
    warnings.warn(
        f"geoformer_amd: GPU_MAX_HW_QUEUES is {'unset (runtime default 4)' if have is None else have}"
        + (" and the HIP runtime is already initialised" if up else "")
        + f": streams will share hardware queues and the sampling / BFS kernels of a forward run one after the other "
        f"(measured: eval forward 183 -> 143 scenes/s, batch-4 training step +3.6 ms).  Export GPU_MAX_HW_QUEUES="
        f"{HW_QUEUES_WANTED} or call geoformer_amd.configure_runtime() before the process's first HIP call"
        + (" (e.g. before torch.cuda.set_device in train.py / test.py)." if up else "."),
        RuntimeWarning, stacklevel=3)


def load():
    """Load the shared library (after ``import torch`` so both share one HIP runtime)."""
    global _lib, EXPORTS
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GeoFormerHipError(
            f"{LIB_PATH} is missing: build it with `python -m geoformer_amd._build` "
            "(or __graft_entry__.build()).  There is no CPU fallback for the HIP operators."
        )
    import torch  # noqa: F401  (loads libamdhip64 first; our library binds to the same SONAME)

    _check_hw_queues(torch)
    lib = ctypes.CDLL(LIB_PATH)
    EXPORTS = _declare(lib)
    if lib.gf_abi_version() != 5:
        raise GeoFormerHipError("libgeoformer_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(status: int, what: str = ""):
    if status != 0:
        msg = load().gf_last_error()
        raise GeoFormerHipError(f"{what} failed ({status}): {msg.decode() if msg else ''}")


FEEDER_MAX_COPIES = 16


class FeederJob(ctypes.Structure):
    """GfFeederJob of include/geoformer_hip.h."""

    _fields_ = [("slot", c_int), ("n_copies", c_int),
                ("src", c_void_p * FEEDER_MAX_COPIES), ("pinned", c_void_p * FEEDER_MAX_COPIES),
                ("dev", c_void_p * FEEDER_MAX_COPIES), ("bytes", c_size_t * FEEDER_MAX_COPIES),
                ("coords_dev", c_void_p), ("N", c_int), ("ncol", c_int), ("mode", c_int), ("pad_", c_int),
                ("scratch", c_void_p), ("input_map", c_void_p), ("head_dev", c_void_p), ("head_host", c_void_p),
                ("stream", c_void_p)]


# seconds the host has spent blocked in the package's own Python-level waits (per process; bench.py's host_busy figure)
host_wait_s = [0.0]


def timed_wait(event):
    """event.synchronize() with its wall time added to host_wait_s."""
    import time

    t = time.perf_counter()
    event.synchronize()
    host_wait_s[0] += time.perf_counter() - t


def ptr(t):
    """Device (or host) address of a tensor as a plain int (ctypes converts it for a c_void_p argtype), None -> NULL."""
    return None if t is None else t.data_ptr()


_raw_stream = None


def stream_ptr():
    """hipStream_t of PyTorch's current stream on the current device (plain int)."""
    global _raw_stream
    import torch

    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
