"""Dev tool (experiment): would the PREVIOUS scene's decoder + mask head fit into the sampling / BFS stretch of the current
one?  The operator calls of one real S150k eval forward are captured (arguments kept) and replayed: the stretch alone
(rest of the sampling on one stream, the BFS on another), the tail alone (4 cross-attention launches + the mask head on a
third stream), and both together."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene, pointops
dev = torch.device("cuda", 0)
batch = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
model = bench.build_model(dev, probe_batch=batch)
NAMES = ("decoder_cross_attn", "mask_head_packed", "mask_head", "geodesic_bfs", "furthest_point_sampling")
calls = []
saved = {n: getattr(pointops, n) for n in NAMES}
def wrap(n, fn):
    def w(*a, **k):
        calls.append((n, a, k))
        return fn(*a, **k)
    return w
for i in range(2):
    np.random.seed(1000 + i)
    with torch.no_grad(): model(batch, 300, training=False)
torch.cuda.synchronize()
for n in NAMES: setattr(pointops, n, wrap(n, saved[n]))
np.random.seed(1002)
with torch.no_grad(): model(batch, 300, training=False)
torch.cuda.synchronize()
for n in NAMES: setattr(pointops, n, saved[n])
print([(n, [tuple(x.shape) if torch.is_tensor(x) else x for x in a][:4], {k: (tuple(v.shape) if torch.is_tensor(v) else v) for k, v in kw.items()}) for n, a, kw in calls])
fps = [c for c in calls if c[0] == "furthest_point_sampling"]
bfs = [c for c in calls if c[0] == "geodesic_bfs"]
tail = [c for c in calls if c[0] in ("decoder_cross_attn", "mask_head_packed", "mask_head")]
s = [torch.cuda.Stream() for _ in range(3)]
def play(cs, st):
    with torch.cuda.stream(st):
        for n, a, k in cs: saved[n](*a, **k)
def run(do_stretch, do_tail):
    main = torch.cuda.current_stream()
    for x in s: x.wait_stream(main)
    if do_stretch:
        play(fps[-1:], s[0])   # the rest of the sampling (the first 256 picks are the earlier call)
        play(bfs, s[1])
    if do_tail: play(tail, s[2])
    for x in s: main.wait_stream(x)
def wall(fn, n=8):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
a = wall(lambda: run(True, False)); b = wall(lambda: run(False, True)); c = wall(lambda: run(True, True))
print("stretch alone %.3f ms   tail alone %.3f ms   together %.3f ms   (sum %.3f)" % (a, b, c, a + b))
