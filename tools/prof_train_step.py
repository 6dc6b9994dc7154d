"""Dev tool: the config-3 training step (batch 4, full epoch) a few times, for rocprofv3 --kernel-trace --stats."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import geoformer_amd
geoformer_amd.configure_runtime()
import numpy as np, torch
from geoformer_amd import scene
from geoformer_amd.model import GeoFormer, InstSetCriterion, load_config
from tests.util import synthetic_state_dict

dev = torch.device("cuda", 0)
mv = lambda d: {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}
cfg = load_config("geoformer_scannet.yaml", batch_size=4, prepare_epochs=120)
m = GeoFormer(cfg); m.load_state_dict(synthetic_state_dict(m.state_dict(), 0)); m.to(dev); m.train()
crit = InstSetCriterion(cfg)
opt = torch.optim.Adam(filter(lambda p: p.requires_grad, m.parameters()), lr=1e-3, fused=os.environ.get("ADAM_FUSED", "1") == "1")
batch = mv(scene.make_batch([scene.make_scene(int(n), 50 + i) for i, n in enumerate((150_000, 120_000, 180_000, 100_000))]))
def step():
    np.random.seed(0)
    out = m(batch, 200)
    loss, _ = crit(out, batch, 200)
    opt.zero_grad(); loss.backward(); opt.step()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for _ in range(2): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize(); print(f"step {(time.perf_counter() - t) / n * 1e3:.1f} ms")
if os.environ.get("PHASES"):
    import collections
    acc = collections.defaultdict(float)
    def tick(name, t0):
        torch.cuda.synchronize(); t1 = time.perf_counter(); acc[name] += t1 - t0; return t1
    orig_bb = m.forward_backbone
    def timed_bb(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = orig_bb(*a, **k)
        tick("fwd backbone", t0)
        return r
    m.forward_backbone = timed_bb
    for _ in range(n):
        np.random.seed(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = m(batch, 200); t1 = tick("fwd total", t0)
        loss, _ = crit(out, batch, 200); t2 = tick("criterion", t1)
        opt.zero_grad(); loss.backward(); t3 = tick("backward", t2)
        opt.step(); tick("optimizer", t3)
    for k, v in acc.items(): print(f"{k:14s} {v / n * 1e3:7.1f} ms")
if os.environ.get("HOSTBOUND"):
    th = tt = 0.0
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        step(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        th += t1 - t0; tt += t2 - t0
    print(f"host returns after {th / n * 1e3:.1f} ms, device done after {tt / n * 1e3:.1f} ms")
if os.environ.get("CPROFILE"):
    import cProfile, pstats, io
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3): step()
    torch.cuda.synchronize()
    pr.disable()
    for key in ("tottime", "cumulative"):
        sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats(key).print_stats(45)
        print(sio.getvalue()[:9000])
if os.environ.get("SYNCS"):
    import traceback, collections
    log = []
    def wrap(owner, name):
        orig = getattr(owner, name)
        def f(*a, **k):
            t0 = time.perf_counter(); r = orig(*a, **k); dt = time.perf_counter() - t0
            if dt > 50e-6:
                fr = [x for x in traceback.extract_stack()[:-1] if "geoformer_amd" in x.filename or "tools/" in x.filename][-1]
                log.append((t0, dt, name, f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"))
            return r
        setattr(owner, name, f)
    for nm in ("tolist", "item", "cpu", "numpy", "__int__", "__bool__", "__float__", "__index__"): wrap(torch.Tensor, nm)
    wrap(torch, "nonzero"); wrap(torch.Tensor, "nonzero"); wrap(torch, "where")
    from geoformer_amd import pointops as _po
    wrap(_po, "legacy_choice"); wrap(np.random, "choice")
    step(); torch.cuda.synchronize(); log.clear()
    T0 = time.perf_counter(); step(); T1 = time.perf_counter(); torch.cuda.synchronize(); T2 = time.perf_counter()
    print(f"host {1e3 * (T1 - T0):.1f} ms, device done {1e3 * (T2 - T0):.1f} ms; blocking / slow host calls:")
    tot = 0
    for t0, dt, nm, where in log:
        print(f"  at {1e3 * (t0 - T0):7.2f} ms  {1e3 * dt:6.2f} ms  {nm:14s} {where}"); tot += dt
    print(f"  total {1e3 * tot:.1f} ms")
if os.environ.get("TPROF"):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for _ in range(2): step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=50, max_name_column_width=60))
if os.environ.get("TPROF_STACK"):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    want = os.environ["TPROF_STACK"].split(",")
    rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key in want]
    rows.sort(key=lambda e: -e.self_cpu_time_total)
    for e in rows[:40]:
        st = [x for x in e.stack if "geoformer_amd" in x or "tools/" in x][:3]
        print(f"{e.key:14s} n {e.count:4d} self {e.self_cpu_time_total / 1e3:7.2f} ms  avg {e.self_cpu_time_total / e.count:6.1f} us  " + " <- ".join(s.split('/')[-1][:60] for s in st))
if os.environ.get("COPIES"):
    import traceback, collections
    cnt = collections.Counter(); byt = collections.Counter()
    def site():
        fr = [x for x in traceback.extract_stack()[:-2] if "geoformer_amd" in x.filename]
        return f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].name}" if fr else "?"
    oc = torch.Tensor.contiguous
    def contiguous(self, *a, **k):
        if not self.is_contiguous():
            s = site(); cnt["contiguous " + s] += 1; byt["contiguous " + s] += self.numel() * self.element_size()
        return oc(self, *a, **k)
    torch.Tensor.contiguous = contiguous
    for nm in ("float", "long", "int", "clone", "to", "half", "double", "bool"):
        def mk(nm, orig):
            def f(self, *a, **k):
                r = orig(self, *a, **k)
                if r.data_ptr() != self.data_ptr() and self.is_cuda:
                    s = site(); cnt[nm + " " + s] += 1; byt[nm + " " + s] += r.numel() * r.element_size()
                return r
            return f
        setattr(torch.Tensor, nm, mk(nm, getattr(torch.Tensor, nm)))
    step(); torch.cuda.synchronize(); cnt.clear(); byt.clear()
    step(); torch.cuda.synchronize()
    print("copies per step by call site (python-visible ones):", sum(cnt.values()))
    for k, v in cnt.most_common(40): print(f"  {v:4d}  {byt[k] / 1e6:9.2f} MB  {k}")
if os.environ.get("OPCOUNT"):
    from torch.profiler import profile, ProfilerActivity
    import collections
    def prof_phase(fn):
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as p:
            r = fn(); torch.cuda.synchronize()
        ev = [e for e in p.events() if e.device_type.name != "CPU"]
        return r, len(ev), sum(e.device_time for e in ev) / 1e3 if ev else 0.0
    np.random.seed(0)
    orig_bb = m.forward_backbone; bbk = [0, 0.0]
    out, nf, tf = prof_phase(lambda: m(batch, 200))
    (loss, _), nc, tc = prof_phase(lambda: crit(out, batch, 200))
    opt.zero_grad()
    _, nb, tb = prof_phase(lambda: loss.backward())
    _, no, to = prof_phase(lambda: opt.step())
    print(f"device activities (kernels + copies): forward {nf} ({tf:.1f} ms), criterion {nc} ({tc:.1f} ms), backward {nb} ({tb:.1f} ms), optimizer {no} ({to:.1f} ms)")
if os.environ.get("BWDNAMES"):
    from torch.profiler import profile, ProfilerActivity
    import collections
    np.random.seed(0)
    which = os.environ["BWDNAMES"]
    if which == "fwd":
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as p:
            out = m(batch, 200); torch.cuda.synchronize()
    else:
        out = m(batch, 200); loss, _ = crit(out, batch, 200); opt.zero_grad()
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as p:
            loss.backward(); torch.cuda.synchronize()
    cnt = collections.Counter(); tim = collections.Counter()
    for e in p.events():
        if e.device_type.name != "CPU":
            k = e.name.split("(")[0][:70]; cnt[k] += 1; tim[k] += e.device_time
    print(which, "launches", sum(cnt.values()))
    for k, v in cnt.most_common(28): print(f"  {v:5d}  {tim[k] / 1e3:7.2f} ms  {k}")
if os.environ.get("NODES"):
    from torch.profiler import profile, ProfilerActivity
    import collections
    np.random.seed(0)
    out = m(batch, 200); loss, _ = crit(out, batch, 200); opt.zero_grad()
    with profile(activities=[ProfilerActivity.CPU]) as p:
        loss.backward(); torch.cuda.synchronize()
    cnt = collections.Counter(); tim = collections.Counter()
    for e in p.key_averages():
        if e.key.startswith("autograd::engine::evaluate_function: "):
            k = e.key.split(": ", 1)[1]; cnt[k] = e.count; tim[k] = e.cpu_time_total
    print("autograd nodes", sum(cnt.values()), "host ms", sum(tim.values()) / 1e3)
    for k, v in sorted(cnt.items(), key=lambda kv: -tim[kv[0]])[:32]: print(f"  {v:5d}  {tim[k] / 1e3:7.2f} ms  {k}")
if os.environ.get("GEMMS"):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as p:
        step(); torch.cuda.synchronize()
    rows = [e for e in p.key_averages(group_by_input_shape=True) if e.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::matmul", "aten::linear")]
    rows.sort(key=lambda e: -e.device_time_total)
    for e in rows[:14]:
        print(f"{e.key:12s} n {e.count:3d} device {e.device_time_total / 1e3:7.2f} ms  avg {e.device_time_total / e.count:8.1f} us  {e.input_shapes}")
if os.environ.get("BFSCAP"):
    # the BFS inputs of the training forward through the -DBFS_TRACE build (GF_LIB_PATH): hops, ring sizes, phases
    from geoformer_amd import pointops, _lib
    from geoformer_amd._lib import ptr, check, stream_ptr
    cap = []
    orig = pointops.geodesic_bfs
    def spy(D, I, deg, src, radius, max_step, wg_threads=1024):
        cap.append((D, I, deg, src.clone(), radius, max_step)); return orig(D, I, deg, src, radius, max_step, wg_threads=wg_threads)
    pointops.geodesic_bfs = spy
    np.random.seed(0); out = m(batch, 200); torch.cuda.synchronize(); pointops.geodesic_bfs = orig
    lib = _lib.load()
    for i, (D, I, deg, src, radius, max_step) in enumerate(cap):
        n, K = D.shape; nq = src.shape[0]
        val = ((D <= radius) & (I >= 0)).sum(1).float()
        for wg in (512, 1024):
            geo = torch.empty((nq, n), dtype=torch.float32, device=dev); keys = torch.empty((nq, n), dtype=torch.int64, device=dev)
            queues = torch.zeros((nq, 10, n), dtype=torch.int32, device=dev)
            for _ in range(2):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.record()
                check(lib.gf_geodesic_bfs_cfg(ptr(D), ptr(I), ptr(deg), n, K, ptr(src), nq, float(radius), int(max_step), ptr(geo), ptr(keys), ptr(queues), queues.numel() // max(nq, 1), wg, stream_ptr()), "bfs")
                e.record(); torch.cuda.synchronize()
            t = queues.view(nq, -1)[:, :24].contiguous().view(torch.int64).cpu().numpy().astype(np.float64)
            hops = np.maximum(t[:, 5], 1)
            print(f"scene {i} n {n} nq {nq} max_step {max_step} wg {wg}: {s.elapsed_time(e) * 1e3:.0f} us; hops {hops.mean():.0f}, ring mean {(t[:, 6] / hops).mean():.0f} max {t[:, 7].max():.0f}; "
                  f"batches/hop {(t[:, 8] / hops).mean():.1f}; in-radius entries per row {val.mean().item():.1f}, rows >16: {(val > 16).float().mean().item():.2f}")
