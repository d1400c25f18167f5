"""Store / copy bandwidth of plain torch kernels at the size of the pooled-level Z matrix (17920 x 1728 fp32 = 124 MB)."""
import torch
def timeit(fn, it=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for mb in (31, 62, 124, 248, 992):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    tz = timeit(lambda: x.zero_()); tc = timeit(lambda: y.copy_(x))
    print("%4d MB: fill %.1f us = %.2f TB/s written; copy %.1f us = %.2f TB/s read + the same written" % (mb, tz * 1e3, mb / tz / 1e3, tc * 1e3, mb / tc / 1e3))
