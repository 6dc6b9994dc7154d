"""Dev experiment: N host threads, each with its own HIP stream and scene, sharing one model."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene
dev = torch.device("cuda", 0)
nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = 40
batches = [bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234 + i)]), dev) for i in range(nthreads)]
model = bench.build_model(dev, probe_batch=batches[0])
def run(i, n):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st), torch.no_grad():
        for _ in range(n):
            model(batches[i], 300, training=False)
    st.synchronize()
for i in range(nthreads):  # warm-up single-threaded (packs weights, caches)
    run(i, 3)
torch.cuda.synchronize()
t = time.perf_counter()
th = [threading.Thread(target=run, args=(i, steps // nthreads)) for i in range(nthreads)]
[x.start() for x in th]; [x.join() for x in th]
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"threads {nthreads}: {steps/dt:.1f} scenes/s  ({dt/steps*1e3:.2f} ms per scene)")
