timeout 300 python -m pytest tests/test_gpu_heads.py tests/test_gpu_model.py -x -q 2>&1 | tail -2
timeout 100 python - <<'PY'
import sys, torch
sys.path.insert(0,'.')
from geoformer_amd import pointops
N, nq = 60108, 256
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
feat, coords, qxyz = r(N, 16), r(N, 3), r(nq, 3)
geo = torch.rand(nq, N, device="cuda", generator=g); geo[geo < 0.3] = -1
mx = torch.rand(nq, device="cuda", generator=g)
w1, b1, w2, b2 = r(nq, 16, 19), r(nq, 16), r(nq, 16), r(nq)
f = lambda: pointops.mask_head(feat, coords, geo, qxyz, mx, w1, b1, w2, b2)
for _ in range(5): f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): f()
e.record(); torch.cuda.synchronize()
print("mask head us", s.elapsed_time(e) / 20 * 1e3)
PY
for rep in 1 2 3; do timeout 150 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c100-170; done
