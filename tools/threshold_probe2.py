"""grid vs scan at small sizes for ball_query, three_nn, knn"""
import ctypes, sys, torch
sys.path.insert(0, ".")
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import sampling
from pytorch_points_amd.ops import knn_points
dev = torch.device("cuda:0")
def knob(name):
    f = getattr(_lib.lib(), name); f.argtypes = [ctypes.c_int]; f.restype = None; return f
bq, tn, kn = knob("pp_debug_set_ball_query_search"), knob("pp_debug_set_three_nn_search"), knob("pp_debug_set_knn_search")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for B, N, M in ((1, 2048, 512), (8, 2048, 512), (32, 2048, 512), (1, 4096, 1024), (8, 4096, 1024), (32, 4096, 1024), (1, 16384, 4096), (4, 16384, 4096)):
    x = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); c = x[:, :: N // M].contiguous()
    r = []
    for m in (0, 1):
        bq(m); r.append(t(lambda: sampling.ball_query(c, x, 0.1, 32)))
    bq(0)
    d2 = torch.empty(B, N, 3, device=dev); idx = torch.empty(B, N, 3, dtype=torch.int32, device=dev)
    for m in (0, 1):
        tn(m); r.append(t(lambda: sampling.three_nn_wrapper(B, N, M, x, c, d2, idx)))
    tn(0)
    for m in (0, 1):
        kn(m); r.append(t(lambda: knn_points(x, x, K=8)))
    kn(0)
    print("B=%-3d N=%-6d M=%-5d  ball grid %6.1f scan %6.1f | three_nn(N unknown, M known) grid %6.1f scan %6.1f | knn8(NxN) grid %6.1f scan %7.1f  (us, eager)" % (B, N, M, *r))
