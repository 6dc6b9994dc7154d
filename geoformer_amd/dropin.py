"""Module-shaped mirrors of the reference's native dependencies, backed by libgeoformer_hip.so.

``install()`` registers them in ``sys.modules`` under the names the UNMODIFIED reference imports:

    spconv, spconv.modules          <- geoformer_amd.spconv           (geoformer.py:4, geoformer_modules.py:2,6)
    PG_OP                           <- lib/pointgroup_ops/src/pointgroup_ops_api.cpp:6-23
    pointnet2, pointnet2._ext       <- lib/pointnet2/_ext_src/src/bindings.cpp:8-21
    faiss, faiss.contrib.torch_utils<- geoformer.py:8-9,172-177; geodesic_utils.py:11-24

Same function names, argument order and in-place/return conventions as the pybind modules, so
``lib/pointgroup_ops/functions/pointgroup_ops.py`` and ``lib/pointnet2/pointnet2_utils.py`` work on top
unchanged.  Bad inputs raise ``RuntimeError`` (the reference's natives ``assert``/``exit(-1)``).
Operators GeoFormer never executes (sec_*, roipool, get_iou, ballquery_batch_p,
bfs_cluster, three_nn / three_interpolate) are bound as well, with the native argument orders.
"""
from __future__ import annotations

import sys
import types

import numpy as np
import torch

from . import pointops, scene


# ------------------------------------------------------------------------------------------
# PG_OP
# ------------------------------------------------------------------------------------------
def voxelize_idx(coords, output_coords, input_map, output_map, batch_size, mode):
    """CPU tensors: fork-safe host path (runs inside DataLoader workers, datasets/scannetv2_inst.py:369), no HIP
    context.  CUDA tensors: the GPU kernels (csrc/voxelize_idx.hip).
    Resizes ``output_coords`` / ``output_map`` like voxelize.cpp:22-26."""
    if coords.is_cuda:
        # device-resident pipeline: same outputs from the GPU kernels, on the input's device
        oc, p2v, v2p = pointops.voxelize_idx(coords.contiguous(), mode)
        output_coords.resize_(oc.shape).copy_(oc)
        input_map.copy_(p2v)
        output_map.resize_(v2p.shape).copy_(v2p)
        return
    oc, p2v, v2p = scene.voxelize_host(coords.numpy(), mode)
    output_coords.resize_(oc.shape).copy_(torch.from_numpy(np.ascontiguousarray(oc)))
    input_map.copy_(torch.from_numpy(p2v))
    output_map.resize_(v2p.shape).copy_(torch.from_numpy(v2p))


def voxelize_fp(feats, output_feats, output_map, mode, nActive, maxActive, nPlane):
    pointops.voxelize_fp(feats, output_map, mode, out=output_feats)


def voxelize_bp(d_output_feats, d_feats, output_map, mode, nActive, maxActive, nPlane):
    pointops.voxelize_bp(d_output_feats, output_map, mode, d_feats)


def point_recover_fp(feats, output_feats, idx_map, nActive, maxActive, nPlane):
    """voxelize.cpp:181-190: the bp kernel with average=false scatters voxel rows back to points."""
    pointops.voxelize_bp(feats, idx_map, 3, output_feats)


def point_recover_bp(d_output_feats, d_feats, idx_map, nActive, maxActive, nPlane):
    pointops.voxelize_fp(d_output_feats, idx_map, 3, out=d_feats)


def _lib_call(name, *args):
    from . import _lib

    _lib.check(getattr(_lib.load(), name)(*args), name)


def _p(t):
    from ._lib import ptr

    return ptr(t)


def _s():
    from ._lib import stream_ptr

    return stream_ptr()


def sec_mean(inp, offsets, out, nProposal, C):
    _lib_call("gf_sec_op", 0, _p(inp), _p(offsets), nProposal, C, _p(out), _s())


def sec_min(inp, offsets, out, nProposal, C):
    _lib_call("gf_sec_op", 1, _p(inp), _p(offsets), nProposal, C, _p(out), _s())


def sec_max(inp, offsets, out, nProposal, C):
    _lib_call("gf_sec_op", 2, _p(inp), _p(offsets), nProposal, C, _p(out), _s())


def roipool_fp(feats, proposals_offset, output_feats, output_maxidx, nProposal, C):
    _lib_call("gf_roipool_fp", _p(feats), _p(proposals_offset), nProposal, C, _p(output_feats), _p(output_maxidx), _s())


def roipool_bp(d_feats, proposals_offset, output_maxidx, d_output_feats, nProposal, C):
    _lib_call("gf_roipool_bp", _p(d_output_feats), _p(output_maxidx), nProposal, C, _p(d_feats), _s())


def get_iou(proposals_idx, proposals_offset, instance_labels, instance_pointnum, proposals_iou, nInstance, nProposal):
    _lib_call("gf_get_iou", _p(proposals_idx), _p(proposals_offset), _p(instance_labels), _p(instance_pointnum),
              nInstance, nProposal, _p(proposals_iou), _s())


def ballquery_batch_p(xyz, batch_idxs, batch_offsets, idx, start_len, n, meanActive, radius):
    """Returns the total pair count like the native (one host sync, bfs_cluster.cu:88)."""
    from . import _lib

    lib = _lib.load()
    scratch = torch.empty(lib.gf_ballquery_batch_p_scratch_bytes(n) // 4 + 4, dtype=torch.int32, device=xyz.device)
    cum = torch.zeros(1, dtype=torch.int32, device=xyz.device)
    _lib_call("gf_ballquery_batch_p", _p(xyz), _p(batch_idxs), _p(batch_offsets), n, meanActive, float(radius), _p(idx),
              _p(start_len), _p(cum), _p(scratch), _s())
    return int(cum.item())


def bfs_cluster(semantic_label, ball_query_idxs, start_len, cluster_idxs, cluster_offsets, N, threshold):
    """CPU tensors in, outputs resized (bfs_cluster.cpp:103-106)."""
    import ctypes

    if semantic_label.is_cuda or ball_query_idxs.is_cuda or start_len.is_cuda:
        raise RuntimeError("bfs_cluster expects CPU tensors")
    ci = torch.zeros((max(N, 1), 2), dtype=torch.int32)
    co = torch.zeros(N + 1, dtype=torch.int32)
    nc, sm = ctypes.c_int32(), ctypes.c_int32()
    _lib_call("gf_bfs_cluster_host", _p(semantic_label.contiguous()), _p(ball_query_idxs.contiguous()),
              _p(start_len.contiguous()), N, threshold, _p(ci), _p(co), ctypes.byref(nc), ctypes.byref(sm))
    cluster_idxs.resize_((sm.value, 2)).copy_(ci[: sm.value])
    cluster_offsets.resize_((nc.value + 1,)).copy_(co[: nc.value + 1])


def three_nn(unknown, known):
    _chk(unknown, torch.float32, "unknown"); _chk(known, torch.float32, "known")
    b, n, _ = unknown.shape
    dist2 = torch.zeros((b, n, 3), dtype=torch.float32, device=unknown.device)
    idx = torch.zeros((b, n, 3), dtype=torch.int32, device=unknown.device)
    _lib_call("gf_three_nn", _p(unknown), _p(known), b, n, known.shape[1], _p(dist2), _p(idx), _s())
    return dist2, idx


def three_interpolate(points, idx, weight):
    _chk(points, torch.float32, "points"); _chk(idx, torch.int32, "idx"); _chk(weight, torch.float32, "weight")
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.zeros((b, c, n), dtype=torch.float32, device=points.device)
    _lib_call("gf_three_interpolate", _p(points), _p(idx), _p(weight), b, c, m, n, _p(out), _s())
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    _chk(grad_out, torch.float32, "grad_out"); _chk(idx, torch.int32, "idx"); _chk(weight, torch.float32, "weight")
    b, c, n = grad_out.shape
    out = torch.zeros((b, c, m), dtype=torch.float32, device=grad_out.device)
    _lib_call("gf_three_interpolate_grad", _p(grad_out), _p(idx), _p(weight), b, c, n, m, _p(out), _s())
    return out


# ------------------------------------------------------------------------------------------
# pointnet2._ext   (callee allocates and returns, sampling.cpp:27-29)
# ------------------------------------------------------------------------------------------
def _chk(t, dtype, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: CPU not supported")  # sampling.cpp:36,62,84
    if not t.is_contiguous() or t.dtype != dtype:
        raise RuntimeError(f"{name} must be a contiguous {dtype} tensor")  # include/utils.h:8-28


def gather_points(points, idx):
    _chk(points, torch.float32, "points"); _chk(idx, torch.int32, "idx")
    return pointops.gather_points(points, idx)


def gather_points_grad(grad_out, idx, n):
    _chk(grad_out, torch.float32, "grad_out"); _chk(idx, torch.int32, "idx")
    return pointops.gather_points_grad(grad_out, idx, n)


def furthest_point_sampling(points, nsamples):
    _chk(points, torch.float32, "points")
    return pointops.furthest_point_sampling(points, nsamples)


def ball_query(new_xyz, xyz, radius, nsample):
    _chk(new_xyz, torch.float32, "new_xyz"); _chk(xyz, torch.float32, "xyz")
    return pointops.ball_query(new_xyz, xyz, radius, nsample)


def group_points(points, idx):
    _chk(points, torch.float32, "points"); _chk(idx, torch.int32, "idx")
    return pointops.group_points(points, idx)


def group_points_grad(grad_out, idx, n):
    _chk(grad_out, torch.float32, "grad_out"); _chk(idx, torch.int32, "idx")
    return pointops.group_points_grad(grad_out, idx, n)


# ------------------------------------------------------------------------------------------
# faiss
# ------------------------------------------------------------------------------------------
class GpuIndexFlatConfig:
    def __init__(self):
        self.useFloat16 = False  # accepted and ignored: distances are exact fp32 (divergence from fp16 faiss)
        self.device = 0


class StandardGpuResources:
    pass


class GpuIndexFlatL2:
    """``GpuIndexFlatL2(res, 3, cfg)`` with ``add / search(x, k, D, I) / reset`` on torch CUDA tensors.
    D receives SQUARED L2 distances, I int64 indices, ordered by (distance, index).  Exact: a grid search
    whose radius doubles until every row holds k neighbours."""

    def __init__(self, res, dim, cfg=None):
        if dim != 3:
            raise RuntimeError("GpuIndexFlatL2 shim: only 3-D points are supported")
        self.base = None

    def add(self, x):
        self.base = x.detach().float().contiguous()

    def reset(self):
        self.base = None

    def search(self, x, k, D, I):
        if self.base is None:
            raise RuntimeError("search() before add()")
        if x.data_ptr() != self.base.data_ptr() and not torch.equal(x, self.base):
            raise NotImplementedError("the shim serves the self-query GeoFormer issues (find_knn, geodesic_utils.py:18)")
        n = self.base.shape[0]
        kk = min(k, n, 64)
        if k > 64:
            raise NotImplementedError("k <= 64")
        span = float((self.base.max(0)[0] - self.base.min(0)[0]).norm().item()) + 1e-3
        radius = 0.05
        while True:
            d2, idx, deg = pointops.knn_radius(self.base, k, radius, sqrt_out=False, check_overflow=True)
            if radius > span or bool((deg + 1 >= kk).all().item()):
                break
            radius *= 1.4
        D.copy_(d2)
        I.copy_(idx.long())
        return D, I


def install():
    """Register the mirrors under the reference's import names.  Idempotent."""
    from . import spconv as sp

    sys.modules["spconv"] = sp
    sys.modules["spconv.modules"] = sp.modules

    pg = types.ModuleType("PG_OP")
    for fn in (voxelize_idx, voxelize_fp, voxelize_bp, point_recover_fp, point_recover_bp, ballquery_batch_p,
               bfs_cluster, roipool_fp, roipool_bp, get_iou, sec_mean, sec_min, sec_max):
        setattr(pg, fn.__name__, fn)
    sys.modules["PG_OP"] = pg

    p2 = types.ModuleType("pointnet2")
    ext = types.ModuleType("pointnet2._ext")
    for fn in (gather_points, gather_points_grad, furthest_point_sampling, ball_query, group_points,
               group_points_grad, three_nn, three_interpolate, three_interpolate_grad):
        setattr(ext, fn.__name__, fn)
    p2._ext = ext
    sys.modules["pointnet2"], sys.modules["pointnet2._ext"] = p2, ext

    fa = types.ModuleType("faiss")
    fa.GpuIndexFlatConfig, fa.StandardGpuResources, fa.GpuIndexFlatL2 = GpuIndexFlatConfig, StandardGpuResources, GpuIndexFlatL2
    fc = types.ModuleType("faiss.contrib")
    ft = types.ModuleType("faiss.contrib.torch_utils")
    fa.contrib, fc.torch_utils = fc, ft
    sys.modules["faiss"], sys.modules["faiss.contrib"], sys.modules["faiss.contrib.torch_utils"] = fa, fc, ft
    # the reference's wrappers allocate with the legacy torch.cuda.FloatTensor(M, C) constructors
    # (pointgroup_ops.py:57,69,93): keep them if this torch still has them, otherwise provide them
    if not hasattr(torch.cuda, "FloatTensor"):
        torch.cuda.FloatTensor = lambda *s: torch.empty(*s, dtype=torch.float32, device="cuda")
        torch.cuda.IntTensor = lambda *s: torch.empty(*s, dtype=torch.int32, device="cuda")
    return {"spconv": sp, "PG_OP": pg, "pointnet2._ext": ext, "faiss": fa}
