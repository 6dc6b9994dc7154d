"""Dev tool: cycle stamps of k_decoder_stage_a's phases (library built with -DDL_TRACE via tools/build_variant.sh and
GF_LIB_PATH): thread 0 of every workgroup, the last stage-A launch of a forward (post of the last layer only) and the one
before it (post + pre)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene, _lib, pointops
dev = torch.device("cuda", 0)
batch = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
m = bench.build_model(dev, probe_batch=batch)
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = torch.zeros(64 * 32, dtype=torch.int64, device="cuda")
A = ["out_mlp+res", "norm3", "linear1", "linear2", "final norm", "norm1+X store", "in_proj qk", "in_proj v"]
Bn = ["self-attention", "out_proj+res", "X store+norm2", "W1 q"]
for i in range(3):
    np.random.seed(1000 + i)
    with torch.no_grad():
        m(batch, 300, training=False)
torch.cuda.synchronize()
raw.gf_dev_dl_trace(ctypes.c_void_p(buf.data_ptr()))
np.random.seed(1003)
with torch.no_grad():
    m(batch, 300, training=False)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 32)
t = t[t[:, 0] != 0]
print("workgroups with stamps:", len(t), "-- stage A: the forward's LAST launch writes words 0..5 (post only), the one before it 6..8; stage B: the last launch")
da = np.array([[w[k] - w[k - 1] for k in range(1, 6)] for w in t])
print("stage A, post part (cycles, median over workgroups):", dict(zip(A[:5], np.median(da, 0).astype(int))), "sum", int(np.median(da.sum(1))))
dp = np.array([[w[7] - w[6], w[8] - w[7]] for w in t])
print("stage A, in_proj (of the launch before):", dict(zip(A[6:], np.median(dp, 0).astype(int))))
db = np.array([[w[k] - w[k - 1] for k in range(17, 21)] for w in t])
print("stage B:", dict(zip(Bn, np.median(db, 0).astype(int))), "sum", int(np.median(db.sum(1))))
