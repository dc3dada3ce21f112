import ctypes, sys, time, torch, numpy as np
sys.path.insert(0, ".")
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import losses, sampling
from pytorch_points_amd.ops import knn_points
dev = torch.device("cuda:0")
mode = _lib.lib().pp_debug_set_nmdistance_search; mode.argtypes = [ctypes.c_int]; mode.restype = None
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for B, N in ((4, 65536), (2, 262144), (64, 4096), (8, 100000)):
    x1 = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); x2 = torch.from_numpy(S.unit_sphere(1, B, N)).to(dev)
    outs = []
    for m in (0, 1):
        mode(m)
        d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
        i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
        ms = t(lambda: losses.nmdistance_forward(x1, x2, d1, d2, i1, i2), 3)
        outs.append((d1, i1, d2, i2, ms))
    mode(0)
    same = all(torch.equal(a, b) for a, b in zip(outs[0][:4], outs[1][:4]))
    print("chamfer fwd B=%d N=%d: grid %.3f ms, brute force %.3f ms, identical=%s" % (B, N, outs[0][4], outs[1][4], same))
    c = x1[:, ::4].contiguous()
    ms = t(lambda: sampling.ball_query(c, x1, 0.05, 32), 3)
    print("   ball_query r=0.05 ns=32 M=%d: %.3f ms" % (c.shape[1], ms))
    ms = t(lambda: knn_points(c, x1, K=8), 3)
    print("   knn K=8 queries=%d: %.3f ms" % (c.shape[1], ms))
