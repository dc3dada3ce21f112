"""grid search vs brute force forward at small sizes: is the N, M >= 2048 switch-over right?"""
import ctypes, sys, torch
sys.path.insert(0, ".")
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import losses
dev = torch.device("cuda:0")
mode = _lib.lib().pp_debug_set_nmdistance_search; mode.argtypes = [ctypes.c_int]; mode.restype = None
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for B, N in ((1, 2048), (8, 2048), (16, 2048), (32, 2048), (64, 2048), (1, 4096), (4, 4096), (8, 4096), (16, 4096), (32, 4096), (1, 8192), (2, 8192), (4, 8192), (1, 16384), (4, 16384), (64, 8192)):
    x1 = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); x2 = torch.from_numpy(S.unit_sphere(1, B, N)).to(dev)
    d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
    i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
    r = []
    for m in (2, 1):
        mode(m)
        r.append(t(lambda: losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)))
    mode(0)
    print("B=%-3d N=%-6d grid %7.1f us   brute force %7.1f us" % (B, N, r[0], r[1]))
