import sys, torch
sys.path.insert(0, ".")
from pytorch_points_amd import synthetic as S
from pytorch_points_amd._ext import losses
dev = torch.device("cuda:0")
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for B, N, C in ((32, 16384, 2), (32, 16384, 6), (32, 16384, 3), (8, 4096, 6)):
    x1 = torch.from_numpy(S.unit_sphere(0, B, N, C)).to(dev); x2 = torch.from_numpy(S.unit_sphere(1, B, N, C)).to(dev)
    d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
    i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
    ms = t(lambda: losses.nmdistance_forward(x1, x2, d1, d2, i1, i2))
    g = torch.rand(B, N, device=dev); o1 = torch.empty_like(x1); o2 = torch.empty_like(x2)
    mb = t(lambda: losses.nmdistance_backward(x1, x2, o1, o2, g, g, i1, i2))
    print("C=%d B=%d N=%d: fwd %.3f ms (%.2f Tpairs/s), bwd %.3f ms" % (C, B, N, ms, 2.0 * B * N * N / ms / 1e9, mb))
