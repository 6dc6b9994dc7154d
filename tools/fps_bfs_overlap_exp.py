"""Dev tool: do furthest point sampling (resumed after 256 picks) and the geodesic BFS overlap on two streams,
for each BFS workgroup size?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import pointops, scene

sc = scene.make_scene(150_000, 1234)
rng = np.random.default_rng(0)
nfg = 60000
pts = sc["xyz"][rng.permutation(sc["xyz"].shape[0])[:nfg]]
xyz = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
sub = xyz[torch.randperm(nfg, device="cuda")[:50000]].contiguous()[None]
D, I, deg = pointops.knn_radius(xyz, 64, 0.05)
first = pointops.furthest_point_sampling(sub, 256)
src = first[0].contiguous()
torch.cuda.synchronize()

def wall(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

print("fps 256 ms", wall(lambda: pointops.furthest_point_sampling(sub, 256)))
print("fps resume 256->2048 ms", wall(lambda: pointops.furthest_point_sampling(sub, 2048, known=first)))
WG = int(os.environ.get("BFS_WG", "1024"))
print("bfs wg_threads", WG, "ms", wall(lambda: pointops.geodesic_bfs(D, I, deg, src, 0.05, 256, wg_threads=WG)))

def both(sa, sb, order="fb"):
    main = torch.cuda.current_stream()
    sa.wait_stream(main); sb.wait_stream(main)
    def f():
        with torch.cuda.stream(sa): pointops.furthest_point_sampling(sub, 2048, known=first)
    def b():
        with torch.cuda.stream(sb): pointops.geodesic_bfs(D, I, deg, src, 0.05, 256, wg_threads=WG)
    (f(), b()) if order == "fb" else (b(), f())
    main.wait_stream(sa); main.wait_stream(sb)

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
print("plain streams, fps then bfs ms", wall(lambda: both(s1, s2)))
print("plain streams, bfs then fps ms", wall(lambda: both(s1, s2, "bf")))

def spans(sa, sb, label):
    main = torch.cuda.current_stream()
    for it in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e0.record(main)
        sa.wait_stream(main); sb.wait_stream(main)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        with torch.cuda.stream(sa):
            ev[0].record(sa); pointops.furthest_point_sampling(sub, 2048, known=first); ev[1].record(sa)
        with torch.cuda.stream(sb):
            ev[2].record(sb); pointops.geodesic_bfs(D, I, deg, src, 0.05, 256, wg_threads=WG); ev[3].record(sb)
        main.wait_stream(sa); main.wait_stream(sb)
        torch.cuda.synchronize()
    print(label, "fps %.2f -> %.2f   bfs %.2f -> %.2f  (ms after the common start)" % tuple(e0.elapsed_time(e) for e in ev))
spans(s1, s2, "issued together:")
