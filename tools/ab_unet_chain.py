"""Dev tool: A/B of the deep U-Net levels as persistent chain launches (gf_dev_unet_chain 1) against one launch per
convolution (0) inside ONE process: eval forwards over the eight benchmark scenes alternate, paired differences; plus
the backbone alone (epoch <= prepare_epochs forward: backbone + semantic head)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from geoformer_amd import scene, _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
batches = [bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234 + i)]), dev) for i in range(8)]
model = bench.build_model(dev, probe_batch=batches[0])
def step(i, ab, epoch):
    lib.gf_dev_unet_chain(ab)
    np.random.seed(1000 + i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.no_grad():
        model(batches[i % 8], epoch, training=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, epoch in (("backbone + semantic head", 0), ("whole forward", 300)):
    for i in range(16): step(i, i % 2, epoch)
    t = {0: [], 1: []}
    for i in range(n):
        for ab in ((0, 1) if (i // 8) % 2 == 0 else (1, 0)):
            t[ab].append(step(i, ab, epoch))
    a, b = np.array(t[0]), np.array(t[1])
    print("%s: separate launches median %.3f ms   chains median %.3f ms   paired diff (chain - separate) median %+.3f ms  mean %+.3f" % (name, np.median(a), np.median(b), np.median(b - a), np.mean(b - a)))
lib.gf_dev_unet_chain(-1)
