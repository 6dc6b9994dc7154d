"""Dev tool: host cost of the native rulebook calls (how launch-bound is the chain?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoformer_amd import sparse, scene, _lib
sc = scene.make_scene(150_000, 1234)
batch = scene.make_batch([sc])
coords = batch["voxel_locs"].int().cuda().contiguous()
shape = tuple(int(s) for s in batch["spatial_shape"])
def host_time(fn, n=20):
    fn(); torch.cuda.synchronize()
    t = 0.0
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t += time.perf_counter() - t0
    return t / n * 1e6
print("down_rules_chain host us (incl. its sync)", host_time(lambda: sparse.down_rules_chain(coords, 1, shape, 6)))
chain = sparse.down_rules_chain(coords, 1, shape, 6)
for l, r in enumerate(chain):
    c = r.out_coords.contiguous()
    print(f"level {l+2}: M {r.M_out}  build_index host us {host_time(lambda: sparse.build_index(c, 1, r.out_shape)):.0f}"
          f"  subm_rules host us {host_time(lambda: sparse.subm_rules(c, r.index_out)):.0f}"
          f"  down_rules host us {host_time(lambda: sparse.down_rules(c, 1, r.out_shape)) if min(r.out_shape) >= 2 else -1:.0f}")
x = torch.zeros(64, device="cuda")
print("torch add_ host us", host_time(lambda: x.add_(1)))
lib = _lib.load()
print("torch.empty host us", host_time(lambda: torch.empty(1000, device="cuda")))
