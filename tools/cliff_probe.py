"""performance cliffs: unusual but legal shapes for the value ops"""
import sys, torch
sys.path.insert(0, ".")
from pytorch_points_amd import synthetic as S
from pytorch_points_amd._ext import sampling
dev = torch.device("cuda:0")
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
B, C = 32, 128
for N, npoint, ns in ((16384, 4096, 64), (16384, 4095, 33), (16383, 4096, 64), (16384, 1023, 33), (20000, 4096, 32), (16384, 4096, 61)):
    idx = torch.randint(0, N, (B, npoint, ns), dtype=torch.int32, device=dev)
    f = torch.randn(B, C, N, device=dev)
    go = torch.randn(B, C, npoint, ns, device=dev)
    a = t(lambda: sampling.group_points(f, idx))
    b = t(lambda: sampling.group_points_grad(go, idx, N))
    gb = 4.0 * B * C * npoint * ns / 1e9
    print("N=%d npoint=%d ns=%d: group_points %.3f ms (%.2f TB/s)  grad %.3f ms" % (N, npoint, ns, a, gb / a, b))
x = torch.from_numpy(S.unit_sphere(0, B, 16384)).to(dev); c = x[:, ::4].contiguous()
for r, ns in ((0.1, 64), (0.1, 128), (0.3, 64), (0.3, 256), (0.6, 32), (0.02, 16)):
    print("ball_query r=%.2f ns=%d: %.3f ms" % (r, ns, t(lambda: sampling.ball_query(c, x, r, ns))))
