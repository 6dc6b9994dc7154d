"""Dev tool: ways to get a 50k-element int64 host array onto the device, host time and completion time."""
import time, numpy as np, torch
n = 50000
arr = np.random.permutation(60108)[:n]
dev = torch.device("cuda")
x = torch.zeros(10, device=dev); torch.cuda.synchronize()
def timeit(name, fn, reps=20):
    fn(); torch.cuda.synchronize()
    th = 0.0; tt = 0.0
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        th += t1 - t0; tt += t2 - t0
    print(f"{name:40s} host {th/reps*1e3:.3f} ms   complete {tt/reps*1e3:.3f} ms")
timeit("torch.tensor(arr, long, device)", lambda: torch.tensor(arr, dtype=torch.long, device=dev))
timeit("from_numpy(arr).to(device)", lambda: torch.from_numpy(arr).to(dev))
timeit("from_numpy(arr).cuda(non_blocking)", lambda: torch.from_numpy(arr).to(dev, non_blocking=True))
pin = torch.empty(1 << 16, dtype=torch.int64).pin_memory()
def via_pin():
    pin[:n].copy_(torch.from_numpy(arr))
    return pin[:n].to(dev, non_blocking=True)
timeit("pinned copy + to(non_blocking)", via_pin)
def via_pin_blocking():
    pin[:n].copy_(torch.from_numpy(arr))
    return pin[:n].to(dev)
timeit("pinned copy + to(blocking)", via_pin_blocking)
a32 = arr.astype(np.int32)
timeit("int32 from_numpy.to(device)", lambda: torch.from_numpy(a32).to(dev))
timeit("int32 -> device -> long", lambda: torch.from_numpy(arr.astype(np.int32)).to(dev).long())
dst = torch.empty(n, dtype=torch.int64, device=dev)
timeit("dst.copy_(from_numpy)", lambda: dst.copy_(torch.from_numpy(arr)))
timeit("dst.copy_(pinned, non_blocking)", lambda: (pin[:n].copy_(torch.from_numpy(arr)), dst.copy_(pin[:n], non_blocking=True)))
ev = torch.cuda.Event()
def with_event():
    ev.synchronize()
    pin[:n].copy_(torch.from_numpy(arr)); out = pin[:n].to(dev, non_blocking=True); ev.record(); return out
timeit("pinned + event", with_event)
