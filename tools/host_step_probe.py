"""Host time (issue only) of the pieces of the eager Chamfer step at config 2"""
import sys, time, torch
sys.path.insert(0, ".")
from pytorch_points_amd import synthetic as S, _lib
from pytorch_points_amd.network.model_loss import nndistance
from pytorch_points_amd._ext import losses
dev = torch.device("cuda:0")
B, N = 32, 16384
x1 = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev).requires_grad_(True)
x2 = torch.from_numpy(S.unit_sphere(1, B, N)).to(dev).requires_grad_(True)
g1 = torch.full((B, N), 1.0 / (B * N), device=dev); g2 = g1.clone()
def measure(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    host = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / n
    return host * 1e6, wall * 1e6
def step():
    x1.grad = None; x2.grad = None
    d1, d2, i1, i2 = nndistance(x1, x2)
    torch.autograd.backward([d1, d2], [g1, g2])
def fwd_grad():
    d1, d2, i1, i2 = nndistance(x1, x2)
def fwd_nograd():
    with torch.no_grad():
        nndistance(x1, x2)
xd1, xd2 = x1.detach(), x2.detach()
o = (torch.empty(B, N, device=dev), torch.empty(B, N, device=dev), torch.empty(B, N, dtype=torch.int32, device=dev), torch.empty(B, N, dtype=torch.int32, device=dev))
gx1 = torch.empty_like(xd1); gx2 = torch.empty_like(xd2)
def ext_fwd(): losses.nmdistance_forward(xd1, xd2, *o)
def ext_bwd(): losses.nmdistance_backward(xd1, xd2, gx1, gx2, g1, g2, o[2], o[3])
for name, fn in (("eager step", step), ("nndistance (grad mode)", fwd_grad), ("nndistance (no_grad)", fwd_nograd), ("ext forward (ctypes)", ext_fwd), ("ext backward (ctypes)", ext_bwd)):
    h, w = measure(fn)
    print("%-26s host %.1f us  wall %.1f us" % (name, h, w))
torch.autograd.set_multithreading_enabled(False)
h, w = measure(step); print("%-26s host %.1f us  wall %.1f us" % ("eager step, calling thread", h, w))
