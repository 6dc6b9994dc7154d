"""Dev tool: where bench.py's secondary.test_py_shape step goes (host timers with a device synchronise behind each part;
the parts overlap in the real loop, so their sum is an upper bound of the step)."""
import os, sys, time


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    import geoformer_amd
    geoformer_amd.configure_runtime()
    import bench
    from geoformer_amd import feeder, postprocess, scene
    dev = torch.device("cuda", 0)
    sizes = np.random.RandomState(7).permutation(np.linspace(0.8, 1.2, 18) * 150_000).astype(int)
    raws = [scene.collate_raw([scene.make_scene(int(n), 7000 + j)]) for j, n in enumerate(sizes)]
    b0 = bench.to_device(scene.make_batch([scene.make_scene(150_000, 1234)]), dev)
    model = bench.build_model(dev, probe_batch=b0)
    THR = float(os.environ.get("THR", "0.0"))
    model.cfg.TEST_SCORE_THRESH = THR
    acc = {}
    def tick(name, t0):
        torch.cuda.synchronize(); acc.setdefault(name, []).append(time.perf_counter() - t0); return time.perf_counter()
    it = iter(feeder.DeviceFeeder(raws, dev))
    for j in range(len(raws)):
        t = time.perf_counter()
        batch = next(it); t = tick("feeder hand-over (finish i, start i+1: staging memcpy, H2D, voxelise)", t)
        np.random.seed(4000 + j)
        with torch.no_grad():
            out = model(batch, 300, training=False); t = tick("forward (proposals read inside)", t)
        cls_final, scores_final, masks_final = out["proposal_scores"]
        if isinstance(cls_final, list):
            continue
        pick = postprocess.matrix_non_max_suppression(masks_final, scores_final, cls_final, final_score_thresh=0.6 * THR); t = tick("matrix NMS", t)
        sel = masks_final[pick]; t = tick("masks_final[pick] (device gather)", t)
        c = sel.cpu().numpy(); t = tick(f"D2H of the picked masks", t)
        scores_final[pick].cpu().numpy(); cls_final[pick].cpu().numpy(); t = tick("D2H scores, classes", t)
        if j == len(raws) - 1:
            print("proposals", tuple(masks_final.shape), masks_final.dtype, "picked", tuple(c.shape), f"{c.nbytes / 1e6:.1f} MB")
    for k, v in acc.items():
        print(f"{k:75s} {np.mean(v[2:]) * 1e3:7.2f} ms  (n={len(v[2:])})")
    # the same scenes resident on the device, same synchronous loop: what the feeder path adds
    res = [bench.to_device(scene.make_batch([scene.make_scene(int(n), 7000 + j)]), dev) for j, n in enumerate(sizes)]
    ts = []
    for j, b in enumerate(res):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        np.random.seed(4000 + j)
        with torch.no_grad():
            model(b, 300, training=False)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{'forward of the same scenes resident in HBM (host voxelisation, untimed)':75s} {np.mean(ts[2:]) * 1e3:7.2f} ms")


if __name__ == "__main__":
    main()
