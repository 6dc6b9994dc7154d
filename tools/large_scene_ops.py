"""Dev tool: where a LARGE scene's forward goes (bench.py's fresh_scenes leg: 192k points cost 7.1 ms against 4.7 for
150k): launch-bound event timings of BFS / sampling / cross-attention / mask head and the forward's wall time per size."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import geoformer_amd
geoformer_amd.configure_runtime()
import numpy as np, torch
import bench
from geoformer_amd import scene, _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
probe = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
m = bench.build_model(dev, probe_batch=probe)
m.reserve_for(250_000)
for n, sd in ((150_000, 1234), (174_852, 5001), (192_169, 5015), (108_214, 5010)):
    b = bench.to_device(scene.make_batch([scene.make_scene(n, sd)]), dev)
    def fwd():
        np.random.seed(7)
        with torch.no_grad():
            return m(b, 300, training=False)
    for _ in range(3): out = fwd()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = fwd()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10 * 1e3
    res = {}
    for op, name in ((0, "bfs"), (3, "fps"), (1, "cross_attn"), (2, "mask_head")):
        ts = []
        for _ in range(3):
            s, e = lib.gf_dev_event_create(), lib.gf_dev_event_create()
            lib.gf_dev_op_kernel_events(op, s, e)
            fwd(); torch.cuda.synchronize()
            if lib.gf_dev_op_kernel_events_taken(op):
                us = ctypes.c_float()
                if lib.gf_dev_event_elapsed_us(s, e, ctypes.byref(us)) == 0: ts.append(us.value)
            else:
                lib.gf_dev_op_kernel_events(op, None, None)
            lib.gf_dev_event_destroy(s); lib.gf_dev_event_destroy(e)
        res[name] = np.mean(ts) if ts else float("nan")
    nfg = int(out["fg_idxs"].shape[0]) if "fg_idxs" in out else -1
    print(f"points {n}: forward {wall:.2f} ms, foreground {nfg}; " + ", ".join(f"{k} {v:.0f} us" for k, v in res.items()), flush=True)
