"""Practical memory-system rates of the box (torch elementwise kernels): copy, fill, read-reduce, at several sizes.
Each measurement is preceded by 200 ms of the same kernel so that it is taken at steady clocks (the first few ms after
idle run ~25 % slower, see bench.py preheat)."""
import time
import torch
dev = torch.device("cuda")
def bench(fn, min_ms=60.0):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 10
    while True:
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        if ms >= min_ms: return ms / n * 1e-3
        n *= 2
for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device=dev, dtype=torch.float32).normal_()
    b = torch.empty_like(a)
    t = bench(lambda: b.copy_(a)); print(f"{mb:5d} MB copy   : {2*mb/1024/t/1e3*1.0737:.2f} TB/s (r+w)")
    t = bench(lambda: b.fill_(1.0)); print(f"{mb:5d} MB fill   : {mb/1024/t/1e3*1.0737:.2f} TB/s (w)")
    t = bench(lambda: a.sum()); print(f"{mb:5d} MB sum    : {mb/1024/t/1e3*1.0737:.2f} TB/s (r)")
