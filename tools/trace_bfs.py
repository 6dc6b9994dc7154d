"""Dev tool: phase cycle stamps of k_geodesic_bfs_lds (variant built with -DBFS_TRACE; first wave of every query) on the
graphs and sources the eval forward itself hands to the BFS on the benchmark scenes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene, pointops, _lib
from geoformer_amd._lib import ptr, check, stream_ptr
lib = _lib.load()
dev = torch.device("cuda", 0)
batches = [bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234 + i)]), dev) for i in range(3)]
model = bench.build_model(dev, probe_batch=batches[0])
cap = []
orig = pointops.geodesic_bfs
def spy(D, I, deg, src, radius, max_step, wg_threads=1024):
    cap.append((D, I, deg, src.clone(), radius, max_step))
    return orig(D, I, deg, src, radius, max_step, wg_threads=wg_threads)
pointops.geodesic_bfs = spy
for i, b in enumerate(batches):
    np.random.seed(1000 + i)
    with torch.no_grad(): model(b, 300, training=False)
torch.cuda.synchronize()
pointops.geodesic_bfs = orig
names = ["expand (to bids issued)", "wait for the atomics", "barrier A", "commit", "barrier B"]
for i, (D, I, deg, src, radius, max_step) in enumerate(cap):
    n, K = D.shape; nq = src.shape[0]
    for wg in (256, 512, 1024):
        geo = torch.empty((nq, n), dtype=torch.float32, device=dev)
        keys = torch.empty((nq, n), dtype=torch.int64, device=dev)
        queues = torch.zeros((nq, 10, n), dtype=torch.int32, device=dev)
        for _ in range(2):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            check(lib.gf_geodesic_bfs_cfg(ptr(D), ptr(I), ptr(deg), n, K, ptr(src), nq, float(radius), int(max_step), ptr(geo), ptr(keys), ptr(queues), wg, stream_ptr()), "bfs")
            e.record(); torch.cuda.synchronize()
        t = queues.view(nq, -1)[:, :24].contiguous().view(torch.int64).cpu().numpy().astype(np.float64)
        hops = t[:, 5]
        tot = t[:, :5].sum(1)
        print(f"scene {i} n {n} wg {wg}: {s.elapsed_time(e) * 1e3:.0f} us; hops {hops.mean():.0f} (max {hops.max():.0f}), ring mean {(t[:, 6] / hops).mean():.0f} max {t[:, 7].max():.0f}; slowest query {tot.max() / tot.mean():.2f} x the mean")
        for j, nme in enumerate(names):
            print(f"     {nme:26s} {(t[:, j] / hops).mean():8.0f} ticks per hop  {100 * (t[:, j] / tot).mean():5.1f} %")
        print(f"     thread 0 per hop: {(t[:, 8] / hops).mean():.2f} batches, {(t[:, 9] / hops).mean():.2f} bids of its group; first batch: "
              f"row entries in registers after {(t[:, 10] / hops).mean():.0f} ticks, bids placed in {(t[:, 11] / hops).mean():.0f}")
    val = ((D <= radius) & (I >= 0)).sum(1).float()
    print(f"     graph: K {K}, entries inside the radius per row: mean {val.mean().item():.1f}, "
          f"rows with more than 16 / 32 / 48: {(val > 16).float().mean().item():.2f} / {(val > 32).float().mean().item():.2f} / {(val > 48).float().mean().item():.2f}")
