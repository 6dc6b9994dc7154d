"""Dev tool: cost of the forward's host RNG draw on this box and whether a helper thread hides it under launches."""
import os, time, threading, sys
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
print("cpus", len(os.sched_getaffinity(0)), "numpy", np.__version__)
n, k = 60108, 50000
def draw(): return np.random.choice(n, k, replace=False)
for _ in range(3): draw()
t = time.perf_counter()
for _ in range(20): draw()
print("draw ms", (time.perf_counter() - t) / 20 * 1e3)
t = time.perf_counter()
for _ in range(20): np.random.permutation(n)
print("permutation ms", (time.perf_counter() - t) / 20 * 1e3)
x = torch.zeros(1000, device="cuda")
def launches(m=25):
    for _ in range(m): x.add_(1.0)
launches(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): launches()
print("25 launches host ms", (time.perf_counter() - t) / 20 * 1e3); torch.cuda.synchronize()
pool = ThreadPoolExecutor(max_workers=1)
pool.submit(draw).result()
def both():
    f = pool.submit(draw)
    launches()
    return f.result()
both()
t = time.perf_counter()
for _ in range(20): both()
print("thread draw + 25 launches ms", (time.perf_counter() - t) / 20 * 1e3); torch.cuda.synchronize()
def serial():
    launches(); return draw()
t = time.perf_counter()
for _ in range(20): serial()
print("serial launches + draw ms", (time.perf_counter() - t) / 20 * 1e3); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): torch.tensor(draw(), dtype=torch.long, device="cuda")
torch.cuda.synchronize()
print("draw + H2D ms", (time.perf_counter() - t) / 20 * 1e3)
