"""Reference point for the write roof: time torch fill / copy of a buffer the size of the batched Hessian."""
import torch
dev = torch.device("cuda:0")
n = 512 * 462 * 462
a = torch.empty(n, dtype=torch.float64, device=dev)
b = torch.empty(n, dtype=torch.float64, device=dev)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
tz = t(lambda: a.zero_())
tf = t(lambda: a.fill_(1.5))
tc = t(lambda: b.copy_(a))
print("bytes %.1f MB  zero_ %.1f us -> %.0f GB/s | fill_ %.1f us -> %.0f GB/s | copy %.1f us -> %.0f GB/s (r+w)" % (
    n * 8 / 1e6, tz * 1e6, n * 8 / tz / 1e9, tf * 1e6, n * 8 / tf / 1e9, tc * 1e6, 2 * n * 8 / tc / 1e9))
