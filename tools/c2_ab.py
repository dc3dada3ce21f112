"""Chamfer forward and forward + backward at config 2 through the extension calls, steady state: A/B of libraries
(PP_LIB=tools/libpp_hip_<tag>.so python tools/c2_ab.py)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("PP_LIB"):
    from pytorch_points_amd import _build
    _build.LIB = os.path.abspath(os.environ["PP_LIB"]); _build.is_stale = lambda: False
from pytorch_points_amd import synthetic as S
from pytorch_points_amd._ext import losses
dev = torch.device("cuda:0")
B, N = 32, 16384
x1 = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); x2 = torch.from_numpy(S.unit_sphere(1, B, N)).to(dev)
d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
g1 = torch.ones(B, N, device=dev); g2 = torch.ones(B, N, device=dev)
gx1 = torch.empty(B, N, 3, device=dev); gx2 = torch.empty(B, N, 3, device=dev)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def fwd(): losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
def step():
    losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
    losses.nmdistance_backward(x1, x2, gx1, gx2, g1, g2, i1, i2)
print("fwd %.4f ms   fwd+bwd %.4f ms   fwd %.4f   fwd+bwd %.4f" % (t(fwd), t(step), t(fwd), t(step)))
