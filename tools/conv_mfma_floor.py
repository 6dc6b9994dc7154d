"""Dev tool (host only, oracle rulebooks): per U-Net level of a benchmark scene, the number of (group, present offset,
16-channel chunk) steps the output-stationary kernels execute and the fp32-MFMA time they imply --
steps x column blocks x 4 v_mfma_f32_16x16x4_f32 (32 cycles each on one SIMD: 157.3 TFLOP/s / 1024 SIMDs / 2.4 GHz =
64 FLOP per cycle) spread over 1024 SIMDs -- beside the algorithmic HBM time at 8 TB/s.
usage: python tools/conv_mfma_floor.py [scene_seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import scene, sparse
from oracle import cpu_backend

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sc = scene.make_scene(150_000, seed)
batch = scene.make_batch([sc])
C = [16 * (i + 1) for i in range(7)]  # geoformer.py:216 (m = 16)
with cpu_backend.installed():
    coords = batch["voxel_locs"].int()
    shape = [int(x) for x in batch["spatial_shape"]]
    rows = []
    lvl = 0
    cur, curshape = coords, shape
    while lvl < 7:
        idx = sparse.build_index(cur, 1, curshape)
        r = sparse.subm_rules(cur, idx)
        nbr = r.nbr.numpy() if torch.is_tensor(r.nbr) else np.asarray(r.nbr)
        M = cur.shape[0]
        K, ld = nbr.shape
        present = nbr[:, :M] >= 0
        pairs = int(present.sum())
        G = (M + 15) // 16
        pad = np.zeros((K, G * 16), bool); pad[:, :M] = present
        gm = pad.reshape(K, G, 16).any(2)
        gsteps = int(gm.sum())
        c = C[lvl]
        nch = ncb = (c + 15) // 16
        mfma = gsteps * nch * ncb * 4
        t_mfma = mfma * 32 / 1024 / 2.4e9 * 1e6
        bytes_ = 4 * (pairs * c + M * c + K * c * c) + 8 * pairs
        t_hbm = bytes_ / 8e12 * 1e6
        rows.append((lvl + 1, M, c, pairs / M, gsteps / G, gsteps * 16 / pairs, t_mfma, t_hbm))
        d = sparse.down_rules(cur, 1, curshape)
        cur = d.out_coords if hasattr(d, "out_coords") else d.coords
        curshape = [(s + 1) // 2 for s in curshape] if not hasattr(d, "out_shape") else list(d.out_shape)
        lvl += 1
print("| level | voxels | C | neighbours / voxel | present offsets / group | tile padding | MFMA floor of a C->C 3x3x3 conv (us) | HBM floor (us) |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.2f}x | {r[6]:.1f} | {r[7]:.1f} |")
