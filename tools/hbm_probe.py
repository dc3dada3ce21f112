"""Practical HBM ceilings on this GPU for the shapes the bandwidth kernels have: pure write (fill),
copy (read+write).  torch kernels, 16 B/lane."""
import torch
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
a = torch.empty(1 << 30, dtype=torch.float32, device=dev)   # 4 GiB
b = torch.empty(1 << 30, dtype=torch.float32, device=dev)
ms = t(lambda: a.fill_(1.0)); print("fill 4 GiB: %.3f ms  %.2f TB/s (write only)" % (ms, 4.295 / ms))
ms = t(lambda: a.zero_()); print("zero 4 GiB: %.3f ms  %.2f TB/s (write only)" % (ms, 4.295 / ms))
ms = t(lambda: b.copy_(a)); print("copy 4 GiB: %.3f ms  %.2f TB/s (read+write)" % (ms, 2 * 4.295 / ms))
ms = t(lambda: torch.sum(a)); print("sum 4 GiB: %.3f ms  %.2f TB/s (read only)" % (ms, 4.295 / ms))
