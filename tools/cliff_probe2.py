"""more cliffs: three_interpolate / gather with unaligned sizes, knn over K, ball_query nsample extremes"""
import sys, torch
sys.path.insert(0, ".")
from pytorch_points_amd import synthetic as S
from pytorch_points_amd._ext import sampling
from pytorch_points_amd.ops import knn_points
dev = torch.device("cuda:0")
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
B, C = 32, 128
for N, M in ((16384, 4096), (16383, 4096), (16384, 4095), (16382, 4099), (50000, 20000)):
    feats = torch.randn(B, C, M, device=dev)
    idx = torch.randint(0, M, (B, N, 3), dtype=torch.int32, device=dev)
    w = torch.rand(B, N, 3, device=dev)
    out = torch.empty(B, C, N, device=dev)
    a = t(lambda: sampling.three_interpolate_wrapper(B, C, M, N, feats, idx, w, out))
    gp = torch.zeros(B, C, M, device=dev)
    b = t(lambda: sampling.three_interpolate_grad_wrapper(B, C, N, M, out, idx, w, gp))
    f2 = torch.randn(B, C, N, device=dev); gi = torch.randint(0, N, (B, M), dtype=torch.int32, device=dev); go = torch.empty(B, C, M, device=dev)
    c = t(lambda: sampling.gather_forward(B, C, N, M, f2, gi, go))
    gg = torch.zeros(B, C, N, device=dev)
    d = t(lambda: sampling.gather_backward(B, C, N, M, go, gi, gg))
    print("N=%d M=%d: three_interpolate %.3f ms, grad %.3f ms | gather fwd %.3f ms, bwd %.3f ms" % (N, M, a, b, c, d))
x = torch.from_numpy(S.unit_sphere(0, B, 16384)).to(dev); cc = x[:, ::4].contiguous()
for K in (1, 2, 4, 8, 16, 24, 32):
    print("knn K=%d (4096 queries x 16384): %.3f ms" % (K, t(lambda: knn_points(cc, x, K=K))))
for ns in (1, 8, 512, 1000):
    print("ball_query r=0.1 ns=%d: %.3f ms" % (ns, t(lambda: sampling.ball_query(cc, x, 0.1, ns))))
