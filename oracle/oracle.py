"""ctypes/numpy front end of the CPU oracle (oracle/gf_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of gf_oracle.c.  Only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import this module; the product package
(geoformer_amd/) never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, byref, c_float, c_int32, c_int64, c_void_p

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libgf_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_DIR, "gf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(src) > os.path.getmtime(_SO):
        subprocess.run(["make", "-C", _DIR, "-B" if force else "-s"], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


# ---- a1/a2 voxelisation ------------------------------------------------------------
def voxelize_idx(coords, mode=4):
    coords = _i64(coords)
    N, ncol = coords.shape
    input_map = np.zeros(N, np.int32)
    M, mx = c_int32(), c_int32()
    oc, om = POINTER(c_int64)(), POINTER(c_int32)()
    L = lib()
    L.orc_voxelize_idx.restype = ctypes.c_int
    rc = L.orc_voxelize_idx(_p(coords), c_int32(N), c_int32(ncol), c_int32(mode), _p(input_map), byref(M), byref(mx),
                            byref(oc), byref(om))
    assert rc == 0
    M, mx = M.value, mx.value
    out_coords = np.ctypeslib.as_array(oc, shape=(max(M * ncol, 1),))[: M * ncol].reshape(M, ncol).copy()
    out_map = np.ctypeslib.as_array(om, shape=(max(M * (mx + 1), 1),))[: M * (mx + 1)].reshape(M, mx + 1).copy()
    L.orc_free(oc)
    L.orc_free(om)
    return out_coords, input_map, out_map


def voxelize_fp(feats, rules, average=True):
    feats, rules = _f32(feats), _i32(rules)
    M, C = rules.shape[0], feats.shape[1]
    out = np.zeros((M, C), np.float32)
    lib().orc_voxelize_fp(_p(feats), _p(rules), c_int32(M), c_int32(rules.shape[1] - 1), c_int32(C),
                          c_int32(int(average)), _p(out))
    return out


def voxelize_bp(d_out, rules, N, average=True):
    d_out, rules = _f32(d_out), _i32(rules)
    M, C = d_out.shape
    d_feats = np.zeros((N, C), np.float32)
    lib().orc_voxelize_bp(_p(d_out), _p(rules), c_int32(M), c_int32(rules.shape[1] - 1), c_int32(C),
                          c_int32(int(average)), _p(d_feats))
    return d_feats


# ---- a4/a5 sparse convolution ------------------------------------------------------
def rules_subm3(coords, shape, ld=None):
    coords = _i32(coords)
    M = coords.shape[0]
    ld = ld or max((M + 15) // 16 * 16, 16)
    nbr = np.full((27, ld), -1, np.int32)
    rc = lib().orc_rules_subm3(_p(coords), c_int32(M), c_int32(shape[0]), c_int32(shape[1]), c_int32(shape[2]),
                               c_int32(ld), _p(nbr))
    assert rc == 0
    return nbr


def rules_down2(coords, shape, ld=None):
    coords = _i32(coords)
    M = coords.shape[0]
    ld = ld or max((M + 15) // 16 * 16, 16)
    out_coords = np.zeros((ld, 4), np.int32)
    child = np.full((8, ld), -1, np.int32)
    parent = np.zeros(max(M, 1), np.int32)
    koff = np.zeros(max(M, 1), np.int32)
    L = lib()
    L.orc_rules_down2.restype = c_int32
    mo = L.orc_rules_down2(_p(coords), c_int32(M), c_int32(shape[0]), c_int32(shape[1]), c_int32(shape[2]),
                           _p(out_coords), c_int32(ld), _p(child), _p(parent), _p(koff))
    return out_coords[:mo].copy(), child, parent[:M], koff[:M]


def conv_fwd(feats, W, nbr, M_out):
    feats, W, nbr = _f32(feats), _f32(W), _i32(nbr)
    K, Cin, Cout = W.shape
    ld = nbr.shape[1]
    out = np.zeros((M_out, Cout), np.float32)
    lib().orc_conv_fwd(_p(feats), _p(W), _p(nbr), c_int32(K), c_int32(M_out), c_int32(ld), c_int32(Cin),
                       c_int32(Cout), _p(out))
    return out


def conv_fwd_f64(feats, W, nbr, M_out):
    """conv_fwd in double precision (the arbiter of the full-size float comparisons)."""
    feats, W, nbr = np.ascontiguousarray(feats, np.float64), np.ascontiguousarray(W, np.float64), _i32(nbr)
    K, Cin, Cout = W.shape
    out = np.zeros((M_out, Cout), np.float64)
    lib().orc_conv_fwd_f64(_p(feats), _p(W), _p(nbr), c_int32(K), c_int32(M_out), c_int32(nbr.shape[1]), c_int32(Cin),
                           c_int32(Cout), _p(out))
    return out


def conv_dgrad(dout, W, nbr, M_in):
    dout, W, nbr = _f32(dout), _f32(W), _i32(nbr)
    K, Cin, Cout = W.shape
    din = np.zeros((M_in, Cin), np.float32)
    lib().orc_conv_dgrad(_p(dout), _p(W), _p(nbr), c_int32(K), c_int32(dout.shape[0]), c_int32(nbr.shape[1]),
                         c_int32(Cin), c_int32(Cout), _p(din))
    return din


def conv_wgrad(feats, dout, nbr, K):
    feats, dout, nbr = _f32(feats), _f32(dout), _i32(nbr)
    Cin, Cout = feats.shape[1], dout.shape[1]
    dW = np.zeros((K, Cin, Cout), np.float32)
    lib().orc_conv_wgrad(_p(feats), _p(dout), _p(nbr), c_int32(K), c_int32(dout.shape[0]), c_int32(nbr.shape[1]),
                         c_int32(Cin), c_int32(Cout), _p(dW))
    return dW


def up_table(parent, koff, ld=None):
    """One-hot table of the inverse conv: up[k,i] = parent[i] iff k == koff[i]."""
    M = parent.shape[0]
    ld = ld or max((M + 15) // 16 * 16, 16)
    up = np.full((8, ld), -1, np.int32)
    up[koff, np.arange(M)] = parent
    return up


# ---- pointnet2 -------------------------------------------------------------------------
def fps(xyz, m):
    xyz = _f32(xyz)
    b, n, _ = xyz.shape
    idx = np.zeros((b, m), np.int32)
    lib().orc_fps(_p(xyz), c_int32(b), c_int32(n), c_int32(m), _p(idx))
    return idx


def gather_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), np.float32)
    lib().orc_gather_points(_p(points), _p(idx), c_int32(b), c_int32(c), c_int32(n), c_int32(m), _p(out))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    b, c, m = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib().orc_gather_points_grad(_p(grad_out), _p(idx), c_int32(b), c_int32(c), c_int32(n), c_int32(m), _p(out))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    new_xyz, xyz = _f32(new_xyz), _f32(xyz)
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    lib().orc_ball_query(_p(new_xyz), _p(xyz), c_int32(b), c_int32(n), c_int32(m), c_float(radius), c_int32(nsample),
                         _p(idx))
    return idx


def group_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, c, n = points.shape
    _, np_, ns = idx.shape
    out = np.zeros((b, c, np_, ns), np.float32)
    lib().orc_group_points(_p(points), _p(idx), c_int32(b), c_int32(c), c_int32(n), c_int32(np_), c_int32(ns), _p(out))
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    b, c, np_, ns = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib().orc_group_points_grad(_p(grad_out), _p(idx), c_int32(b), c_int32(c), c_int32(n), c_int32(np_), c_int32(ns),
                                _p(out))
    return out


def three_nn(unknown, known):
    unknown, known = _f32(unknown), _f32(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    lib().orc_three_nn(_p(unknown), _p(known), c_int32(b), c_int32(n), c_int32(m), _p(d2), _p(idx))
    return d2, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), np.float32)
    lib().orc_three_interpolate(_p(points), _p(idx), _p(weight), c_int32(b), c_int32(c), c_int32(m), c_int32(n),
                                _p(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, idx, weight = _f32(grad_out), _i32(idx), _f32(weight)
    b, c, n = grad_out.shape
    out = np.zeros((b, c, m), np.float32)
    lib().orc_three_interpolate_grad(_p(grad_out), _p(idx), _p(weight), c_int32(b), c_int32(c), c_int32(n),
                                     c_int32(m), _p(out))
    return out


# ---- kNN + geodesic ------------------------------------------------------------------------
def knn(base, query, k):
    base, query = _f32(base), _f32(query)
    n, nq = base.shape[0], query.shape[0]
    D = np.zeros((nq, k), np.float32)
    I = np.zeros((nq, k), np.int64)
    lib().orc_knn(_p(base), c_int32(n), _p(query), c_int32(nq), c_int32(k), _p(D), _p(I))
    return D, I


def geodesic(dist_arr, idx_arr, query_inds, radius, max_step):
    """dist_arr/idx_arr: [n, k-1] (self column dropped, distances sqrt'ed)."""
    dist_arr, idx_arr, query_inds = _f32(dist_arr), _i64(idx_arr), _i64(query_inds)
    n, kk = dist_arr.shape
    nq = query_inds.shape[0]
    geo = np.zeros((nq, n), np.float32)
    lib().orc_geodesic(_p(dist_arr), _p(idx_arr), c_int32(n), c_int32(kk), _p(query_inds), c_int32(nq),
                       c_float(radius), c_int32(max_step), _p(geo))
    return geo


def sec_op(kind, inp, offsets):
    inp, offsets = _f32(inp), _i32(offsets)
    nP, C = offsets.shape[0] - 1, inp.shape[1]
    out = np.zeros((nP, C), np.float32)
    getattr(lib(), f"orc_sec_{kind}")(_p(inp), _p(offsets), c_int32(nP), c_int32(C), _p(out))
    return out


def roipool_fp(feats, offsets):
    feats, offsets = _f32(feats), _i32(offsets)
    nP, C = offsets.shape[0] - 1, feats.shape[1]
    out = np.zeros((nP, C), np.float32)
    arg = np.zeros((nP, C), np.int32)
    lib().orc_roipool_fp(_p(feats), _p(offsets), c_int32(nP), c_int32(C), _p(out), _p(arg))
    return out, arg


def get_iou(pidx, poff, inst_labels, inst_pointnum):
    pidx, poff, inst_labels, inst_pointnum = _i32(pidx), _i32(poff), _i64(inst_labels), _i32(inst_pointnum)
    nP, nI = poff.shape[0] - 1, inst_pointnum.shape[0]
    iou = np.zeros((nP, nI), np.float32)
    lib().orc_get_iou(_p(pidx), _p(poff), _p(inst_labels), _p(inst_pointnum), c_int32(nI), c_int32(nP), _p(iou))
    return iou


def ballquery_batch_p(xyz, batch_idxs, batch_offsets, mean_active, radius):
    xyz, batch_idxs, batch_offsets = _f32(xyz), _i32(batch_idxs), _i32(batch_offsets)
    n = xyz.shape[0]
    idx = np.zeros(n * mean_active, np.int32)
    start_len = np.zeros((n, 2), np.int32)
    L = lib()
    L.orc_ballquery_batch_p.restype = c_int32
    cum = L.orc_ballquery_batch_p(_p(xyz), _p(batch_idxs), _p(batch_offsets), c_int32(n), c_int32(mean_active),
                                  c_float(radius), _p(idx), _p(start_len))
    return cum, idx, start_len


def bfs_cluster(sem, bq_idx, start_len, threshold):
    sem, bq_idx, start_len = _i32(sem), _i32(bq_idx), _i32(start_len)
    N = sem.shape[0]
    ci = np.zeros((N, 2), np.int32)
    co = np.zeros(N + 1, np.int32)
    nc, sm = c_int32(), c_int32()
    lib().orc_bfs_cluster(_p(sem), _p(bq_idx), _p(start_len), c_int32(N), c_int32(threshold), _p(ci), _p(co), byref(nc),
                          byref(sm))
    return ci[: sm.value].copy(), co[: nc.value + 1].copy()


def proposal_stats(mask_logits, cls_logits, sem_prob, logit_thresh, score_thresh, npoint_thresh, min_class=4):
    mask_logits, cls_logits, sem_prob = _f32(mask_logits), _f32(cls_logits), _f32(sem_prob)
    nq, N = mask_logits.shape
    ncls = cls_logits.shape[1]
    cls_pred, npts, fin = np.zeros(nq, np.int32), np.zeros(nq, np.int32), np.zeros(nq, np.int32)
    scores = np.zeros(nq, np.float32)
    lib().orc_proposal_stats(_p(mask_logits), _p(cls_logits), _p(sem_prob), c_int32(nq), c_int32(N), c_int32(ncls),
                             c_float(logit_thresh), c_float(score_thresh), c_int32(npoint_thresh), c_int32(min_class),
                             _p(cls_pred), _p(npts), _p(scores), _p(fin))
    return cls_pred, npts, scores, fin


def proposal_stats_fs(mask_logits, sim, logit_thresh, score_thresh, npoint_thresh, sim_thresh):
    mask_logits, sim = _f32(mask_logits), _f32(sim)
    nq, N = mask_logits.shape
    npts, fin = np.zeros(nq, np.int32), np.zeros(nq, np.int32)
    scores = np.zeros(nq, np.float32)
    lib().orc_proposal_stats_fs(_p(mask_logits), _p(sim), c_int32(nq), c_int32(N), c_float(logit_thresh),
                                c_float(score_thresh), c_int32(npoint_thresh), c_float(sim_thresh), _p(npts), _p(scores),
                                _p(fin))
    return npts, scores, fin


def proposal_scatter(mask_logits, sel, fg_idxs, logit_thresh, num_points):
    mask_logits, sel, fg_idxs = _f32(mask_logits), _i32(sel), _i64(fg_idxs)
    out = np.zeros((sel.shape[0], num_points), np.int32)
    lib().orc_proposal_scatter(_p(mask_logits), _p(sel), c_int32(sel.shape[0]), c_int32(mask_logits.shape[1]),
                               _p(fg_idxs), c_float(logit_thresh), c_int32(num_points), _p(out))
    return out


def mask_intersections(masks):
    masks = _i32(masks)
    n, N = masks.shape
    inter = np.zeros((n, n), np.int32)
    lib().orc_mask_intersections(_p(masks), c_int32(n), c_int32(N), _p(inter))
    return inter
