"""Shared helpers for the parity tests (seeded synthetic voxel sets)."""
import numpy as np


def random_voxels(rng, M, shape, batch=1, surface=True):
    """Unique int32 (b,x,y,z) rows in random (non-sorted) order."""
    X, Y, Z = shape
    rows = []
    for b in range(batch):
        if surface:  # a few axis-aligned sheets: neighbour statistics like a scanned room
            pts = []
            per = max(M // batch // 3, 1)
            for axis in range(3):
                c = rng.integers(0, shape[axis])
                p = np.stack([rng.integers(0, X, per * 2), rng.integers(0, Y, per * 2), rng.integers(0, Z, per * 2)], 1)
                p[:, axis] = c + rng.integers(0, 2, per * 2)
                p[:, axis] = np.clip(p[:, axis], 0, shape[axis] - 1)
                pts.append(p)
            p = np.concatenate(pts)
        else:
            p = np.stack([rng.integers(0, X, M * 2), rng.integers(0, Y, M * 2), rng.integers(0, Z, M * 2)], 1)
        p = np.unique(p, axis=0)
        rng.shuffle(p)
        p = p[: max(M // batch, 1)]
        rows.append(np.concatenate([np.full((p.shape[0], 1), b), p], 1))
    return np.concatenate(rows).astype(np.int32)
