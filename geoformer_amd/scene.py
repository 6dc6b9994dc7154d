"""Synthetic ScanNet-like scenes and the reference batch-dict schema.

No dataset ships with the reference (``.gitignore:66,68``) and there is no network, so the
benchmarks and parity tests run on generated rooms built to SURVEY.md §8(d): a room box
(floor + 4 walls, no ceiling) with axis-aligned cuboids on the floor, every rectangle
sampled on a jittered lattice so the point spacing (~2 cm at 150k points) and therefore
the voxel occupancy statistics (≈0.94 voxels/point, 6-12 submanifold taps per voxel)
resemble ``_vh_clean_2`` meshes.  The batch dict mirrors what
``datasets/scannetv2_inst.py:371-387`` hands to ``GeoFormer.forward``.
"""
from __future__ import annotations

import numpy as np


def _rect_points(rng, origin, eu, ev, normal, spacing):
    """Jittered lattice on the rectangle origin + a*eu + b*ev, a,b in [0,1]."""
    lu, lv = np.linalg.norm(eu), np.linalg.norm(ev)
    nu, nv = max(int(round(lu / spacing)), 1), max(int(round(lv / spacing)), 1)
    a = (np.arange(nu) + 0.5) / nu
    b = (np.arange(nv) + 0.5) / nv
    A, Bm = np.meshgrid(a, b, indexing="ij")
    A = A + rng.uniform(-0.25, 0.25, A.shape) / nu
    Bm = Bm + rng.uniform(-0.25, 0.25, Bm.shape) / nv
    pts = origin[None, :] + A.reshape(-1, 1) * eu[None, :] + Bm.reshape(-1, 1) * ev[None, :]
    pts = pts + rng.normal(0.0, 0.001, (pts.shape[0], 1)) * normal[None, :]
    return pts


def make_scene(n_points: int = 150_000, seed: int = 1234, room=None, n_boxes=None):
    """Returns dict(xyz float32 [N,3] (mean-centred metres), rgb float32 [N,3] in [-1,1],
    label int64 [N] (0 floor, 1 wall, 4..12 objects), instance int64 [N] (-100 or id))."""
    rng = np.random.default_rng(seed)
    if room is None:
        W, D, H = rng.uniform(4, 6), rng.uniform(3, 4.5), 2.5
    else:
        W, D, H = room
    if n_boxes is None:
        n_boxes = int(rng.integers(6, 13))
    rects = []  # (origin, eu, ev, normal, label, instance)
    ex, ey, ez = np.eye(3)
    rects.append((np.array([-W / 2, -D / 2, 0.0]), W * ex, D * ey, ez, 0, -100))
    rects.append((np.array([-W / 2, -D / 2, 0.0]), W * ex, H * ez, ey, 1, -100))
    rects.append((np.array([-W / 2, D / 2, 0.0]), W * ex, H * ez, ey, 1, -100))
    rects.append((np.array([-W / 2, -D / 2, 0.0]), D * ey, H * ez, ex, 1, -100))
    rects.append((np.array([W / 2, -D / 2, 0.0]), D * ey, H * ez, ex, 1, -100))
    for i in range(n_boxes):
        sz = rng.uniform(0.3, 1.2, 3)
        sz[0], sz[1] = min(sz[0], W * 0.45), min(sz[1], D * 0.45)
        cx = rng.uniform(-W / 2 + sz[0] / 2, W / 2 - sz[0] / 2)
        cy = rng.uniform(-D / 2 + sz[1] / 2, D / 2 - sz[1] / 2)
        o = np.array([cx - sz[0] / 2, cy - sz[1] / 2, 0.0])
        lab, inst = 4 + (i % 9), i
        rects.append((o + sz[2] * ez, sz[0] * ex, sz[1] * ey, ez, lab, inst))  # top
        rects.append((o, sz[0] * ex, sz[2] * ez, ey, lab, inst))
        rects.append((o + sz[1] * ey, sz[0] * ex, sz[2] * ez, ey, lab, inst))
        rects.append((o, sz[1] * ey, sz[2] * ez, ex, lab, inst))
        rects.append((o + sz[0] * ex, sz[1] * ey, sz[2] * ez, ex, lab, inst))
    area = sum(np.linalg.norm(r[1]) * np.linalg.norm(r[2]) for r in rects)
    spacing = float(np.sqrt(area / n_points))
    xyz, rgb, lab, inst = [], [], [], []
    for (o, eu, ev, nrm, l, ins) in rects:
        p = _rect_points(rng, o, eu, ev, nrm, spacing)
        xyz.append(p)
        base = rng.uniform(-1, 1, 3)
        rgb.append(np.clip(base[None, :] + rng.normal(0, 0.05, (p.shape[0], 3)), -1, 1))
        lab.append(np.full(p.shape[0], l, np.int64))
        inst.append(np.full(p.shape[0], ins, np.int64))
    xyz = np.concatenate(xyz).astype(np.float64)
    xyz -= xyz.mean(0, keepdims=True)  # data/scannetv2/prepare_data_inst.py:44 mean-centres
    return {
        "xyz": xyz.astype(np.float32),
        "rgb": np.concatenate(rgb).astype(np.float32),
        "label": np.concatenate(lab),
        "instance": np.concatenate(inst),
        "spacing": spacing,
    }


def make_small_scene(n_points: int = 8192, seed: int = 7):
    """S8k: a 1.2 x 1.0 x 0.8 m box on a 1.6 x 1.6 m floor patch at ScanNet density."""
    return make_scene(n_points, seed, room=(1.6, 1.6, 0.6), n_boxes=1)


def voxelize_host(locs: np.ndarray, mode: int = 4):
    """Host voxelisation with the semantics of PG_OP.voxelize_idx (first-occurrence voxel
    order, rule rows [count, point ids..., 0 pad]; lib/pointgroup_ops/src/voxelize/
    voxelize.cpp:58-152) in vectorised numpy.  Product code for the synthetic-data
    harness (DataLoader side of the boundary); checked against the oracle in tests."""
    N = locs.shape[0]
    ext = locs.max(0).astype(np.int64) + 1 if N else np.ones(locs.shape[1], np.int64)
    if N and locs.min() >= 0 and float(np.prod(ext.astype(np.float64))) < 2.0 ** 62:
        key = np.zeros(N, np.int64)  # one packed integer per row: np.unique on it is ~50x faster than on records
        for c in range(locs.shape[1]):
            key = key * ext[c] + locs[:, c]
    else:
        key = np.ascontiguousarray(locs).view([("", locs.dtype)] * locs.shape[1]).reshape(N)
    _, first, inv, counts = np.unique(key, return_index=True, return_inverse=True, return_counts=True)
    order = np.argsort(first, kind="stable")  # voxel ids in order of first occurrence
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    p2v = rank[inv.reshape(-1)].astype(np.int32)
    M = order.size
    cnt = counts[order]
    max_active = int(cnt.max()) if (mode in (3, 4) and M > 0) else 1
    v2p = np.zeros((M, max_active + 1), np.int32)
    pts_sorted = np.argsort(p2v, kind="stable")  # points grouped by voxel, ascending point id inside
    starts = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    within = np.arange(N) - np.repeat(starts, cnt)
    vox_of = p2v[pts_sorted]
    if mode in (3, 4):
        v2p[:, 0] = cnt
        v2p[vox_of, 1 + within] = pts_sorted
    elif mode == 2:
        v2p[:, 0] = 1
        v2p[:, 1] = pts_sorted[starts + cnt - 1]
    else:
        v2p[:, 0] = 1
        v2p[:, 1] = pts_sorted[starts]
    voxel_locs = locs[first[order]]
    return voxel_locs, p2v, v2p


def collate_raw(scenes, scale: int = 50, full_scale_min: int = 128):
    """The collate step of ``datasets/scannetv2_inst.py:testMerge/trainMerge`` without the voxelisation: per-point
    tensors, offsets, spatial shape.  ``scenes`` is a list of make_scene().  ``geoformer_amd.feeder.DeviceFeeder``
    uploads such a dict and voxelises it on the GPU; ``make_batch`` voxelises on the host."""
    import torch

    locs, locs_float, feats, labels, insts, offsets, mins, maxs = [], [], [], [], [], [0], [], []
    inst_base = 0
    for b, sc in enumerate(scenes):
        xyz_middle = sc["xyz"].astype(np.float64)
        xyz = xyz_middle * scale
        xyz = xyz - xyz.min(0)
        n = xyz.shape[0]
        loc = np.concatenate([np.full((n, 1), b, np.int64), xyz.astype(np.int64)], 1)
        locs.append(loc)
        locs_float.append(xyz_middle.astype(np.float32))
        feats.append(sc["rgb"].astype(np.float32))
        labels.append(sc["label"])
        ins = sc["instance"].copy()
        ins[ins >= 0] += inst_base
        inst_base += int(sc["instance"].max()) + 1 if (sc["instance"] >= 0).any() else 0
        insts.append(ins)
        offsets.append(offsets[-1] + n)
        mins.append(xyz_middle.min(0).astype(np.float32))
        maxs.append(xyz_middle.max(0).astype(np.float32))
    locs = np.concatenate(locs)
    spatial_shape = np.clip(locs.max(0)[1:] + 1, full_scale_min, None)
    return {
        "locs": torch.from_numpy(locs),
        "locs_float": torch.from_numpy(np.concatenate(locs_float)),
        "feats": torch.from_numpy(np.concatenate(feats)),
        "labels": torch.from_numpy(np.concatenate(labels)),
        "instance_labels": torch.from_numpy(np.concatenate(insts)),
        "offsets": torch.tensor(offsets, dtype=torch.int32),
        "spatial_shape": spatial_shape,
        "pc_mins": torch.from_numpy(np.stack(mins)),
        "pc_maxs": torch.from_numpy(np.stack(maxs)),
    }


def make_batch(scenes, scale: int = 50, full_scale_min: int = 128, mode: int = 4):
    """Batch dict as built by ``datasets/scannetv2_inst.py:testMerge/trainMerge`` (CPU numpy
    arrays / torch tensors are produced by the caller).  ``scenes`` is a list of make_scene()."""
    import torch

    batch = collate_raw(scenes, scale, full_scale_min)
    voxel_locs, p2v, v2p = voxelize_host(batch["locs"].numpy(), mode)
    batch["voxel_locs"] = torch.from_numpy(voxel_locs)
    batch["p2v_map"] = torch.from_numpy(p2v)
    batch["v2p_map"] = torch.from_numpy(v2p)
    return batch
