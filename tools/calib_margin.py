"""Dev tool: the calibrated 1e-4 full-forward comparison (tests/test_gpu_fullsize.py) for several calibration thread
counts: stage maxima of |GPU - host| per count (how much margin the bound has under each calibrated state)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import test_gpu_fullsize as T
from tests.util import calibrated_benchmark_state
from geoformer_amd import scene

pts = 150_000
sc = scene.make_scene(pts, 1234)
batch = scene.make_batch([sc])
s150k = (sc, batch, None, None)
for th in [int(a) for a in sys.argv[1:]] or [1, 8]:
    t0 = time.time()
    state, _ = calibrated_benchmark_state(batch, threads=th)
    t1 = time.time()
    seen = []
    def close_abs(got, ref):
        seen.append(float(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).max()))
        return True
    try:
        info = T._forward_gpu_against_host(s150k, close_abs, state=state)
    except AssertionError as e:
        info = "assertion: " + str(e)[:200]
    print(f"threads {th}: calibration {t1 - t0:.1f} s, comparison {time.time() - t1:.1f} s, max {max(seen):.3e}, stages",
          " ".join(f"{d:.2e}" for d in seen), info, flush=True)
