"""Dev tool: time FPS / ball query / kNN graph / BFS at S150k-like sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import pointops, scene

def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

sc = scene.make_scene(150_000, 1234)
rng = np.random.default_rng(0)
nfg = 60000
pts = sc["xyz"][rng.permutation(sc["xyz"].shape[0])[:nfg]]
xyz = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
sub = xyz[torch.randperm(nfg, device="cuda")[:50000]].contiguous()[None]
print("fps 50000->2048 us", timeit(lambda: pointops.furthest_point_sampling(sub, 2048)))
idx = pointops.furthest_point_sampling(sub, 2048)
ctr = sub[0][idx[0].long()][None].contiguous()
print("ball query us", timeit(lambda: pointops.ball_query(ctr, sub, 0.2, 64)))
bq = pointops.ball_query(ctr, sub, 0.2, 64)
f = torch.randn(1, 16, 50000, device="cuda")
print("group c=16 us", timeit(lambda: pointops.group_points(f, bq)))
print("knn graph us", timeit(lambda: pointops.knn_radius(xyz, 64, 0.05)))
D, I, deg = pointops.knn_radius(xyz, 64, 0.05, check_overflow=True)
print("deg mean", deg.float().mean().item(), "max", deg.max().item())
src = idx[0, :256].contiguous()
print("bfs nq=256 us", timeit(lambda: pointops.geodesic_bfs(D, I, deg, src, 0.05, 256)))
geo = pointops.geodesic_bfs(D, I, deg, src, 0.05, 256)
print("reached frac", (geo >= 0).float().mean().item(), "max geo", geo.max().item())
if os.environ.get("BFS_STEPS"):
    for ms in (0, 16, 64, 128, 192, 256, 512):
        t = timeit(lambda: pointops.geodesic_bfs(D, I, deg, src, 0.05, ms))
        gg = pointops.geodesic_bfs(D, I, deg, src, 0.05, ms)
        print(f"bfs max_step={ms}: {t:.1f} us, reached {(gg >= 0).float().mean().item():.4f}")
