"""Dev tool: scenes/s of an eval loop that gets every scene from the host -- blocking path of the reference's drivers
(host voxelisation in the collate + .cuda() per tensor) against geoformer_amd.feeder.DeviceFeeder (pinned staging,
copy stream, GPU voxelisation one batch ahead).  The collate itself (numpy concatenations) is done up front for both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene
from geoformer_amd.feeder import DeviceFeeder
dev = torch.device("cuda", 0)
scenes = [scene.make_scene(150_000, 1234 + i) for i in range(6)]
raws = [scene.collate_raw([s]) for s in scenes]
model = bench.build_model(dev, probe_batch=bench.to_device(scene.make_batch([scenes[0]]), dev))
K = 24
def fwd(b, i):
    np.random.seed(i)
    with torch.no_grad():
        return model(b, 300, training=False)
def blocking():
    for i in range(K):
        raw = dict(raws[i % len(raws)])
        vl, p2v, v2p = scene.voxelize_host(raw["locs"].numpy(), 4)
        raw["voxel_locs"], raw["p2v_map"], raw["v2p_map"] = torch.from_numpy(vl), torch.from_numpy(p2v), torch.from_numpy(v2p)
        fwd(bench.to_device(raw, dev), i)
def fed():
    for i, b in enumerate(DeviceFeeder((raws[i % len(raws)] for i in range(K)), dev)):
        fwd(b, i)
for name, fn in (("blocking (host voxelisation + .cuda())", blocking), ("DeviceFeeder", fed), ("blocking", blocking), ("DeviceFeeder", fed)):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"{name:42s} {K / dt:7.1f} scenes/s  {dt / K * 1e3:6.2f} ms per scene")
