"""Dev tool: does the latency-bound stage (FPS 256->2048 beside the geodesic BFS) of TWO scenes run concurrently in
about the time of one?  Four streams: FPS_A, BFS_A, FPS_B, BFS_B."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoformer_amd import pointops, scene

def prep(seed, nfg):
    sc = scene.make_scene(150_000, seed)
    rng = np.random.default_rng(seed)
    pts = sc["xyz"][rng.permutation(sc["xyz"].shape[0])[:nfg]]
    xyz = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
    sub = xyz[torch.randperm(nfg, device="cuda")[:50000]].contiguous()[None]
    D, I, deg = pointops.knn_radius(xyz, 64, 0.05)
    first = pointops.furthest_point_sampling(sub, 256)
    return sub, D, I, deg, first, first[0].contiguous()

A, B = prep(1234, 60000), prep(1235, 64000)
torch.cuda.synchronize()
ss = [torch.cuda.Stream() for _ in range(4)]

def stage(S, sf, sb):
    sub, D, I, deg, first, src = S
    with torch.cuda.stream(sf): pointops.furthest_point_sampling(sub, 2048, known=first)
    with torch.cuda.stream(sb): pointops.geodesic_bfs(D, I, deg, src, 0.05, 256, wg_threads=256)

def run(two):
    main = torch.cuda.current_stream()
    for s in ss: s.wait_stream(main)
    stage(A, ss[0], ss[1])
    if two: stage(B, ss[2], ss[3])
    for s in ss: main.wait_stream(s)

def wall(fn, n=6):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

print("one scene  (FPS rest || BFS): %.3f ms" % wall(lambda: run(False)))
print("two scenes (4 kernels at once): %.3f ms  -> %.3f ms per scene" % ((lambda t: (t, t / 2))(wall(lambda: run(True)))))
def seq():
    run(False)
    main = torch.cuda.current_stream()
    for s in ss: s.wait_stream(main)
    stage(B, ss[2], ss[3])
    for s in ss: main.wait_stream(s)
print("two scenes one after the other: %.3f ms" % wall(seq))
