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


def synthetic_state_dict(ref_sd, seed=0):
    """Deterministic weights keyed by parameter NAME (not by construction order), so the reference
    model (golden generator) and the build's model (GPU test) get identical values without a 32 MB
    checkpoint in the repo.  BatchNorm statistics are randomised so the BN path is exercised."""
    import zlib

    import torch

    out = {}
    for name, t in ref_sd.items():
        rng = np.random.default_rng((zlib.crc32(name.encode()) + seed * 7919) % (2**32))
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            v = np.array(100, dtype=np.int64)
        elif name.endswith("running_mean"):
            v = rng.normal(0, 0.1, shape)
        elif name.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, shape)
        elif name.endswith("gauss_B"):
            v = rng.normal(0, 1.0, shape)
        elif name.endswith(".alpha") or (t.dim() == 1 and name.endswith("weight")):
            v = rng.uniform(0.5, 1.5, shape)  # norm scales
        elif t.dim() == 1:
            v = rng.normal(0, 0.1, shape)  # biases
        else:
            if t.dim() == 5:  # sparse conv [k,k,k,Cin,Cout]: every tap contributes
                fan_in = shape[3] * max(shape[0] * shape[1] * shape[2] // 3, 1)
            else:
                fan_in = int(np.prod(shape[1:]))
            v = rng.normal(0, 1.0 / np.sqrt(max(fan_in, 1)), shape)
        out[name] = torch.from_numpy(np.asarray(v)).to(t.dtype)
    return out
