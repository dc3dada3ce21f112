"""group_points_grad at config 4 (B=32, C=128, N=16384, npoint=4096, nsample=64): time, and the result against the default library's"""
import os, sys, torch
sys.path.insert(0, ".")
if os.environ.get("PP_LIB"):
    from pytorch_points_amd import _build
    _build.LIB = os.path.abspath(os.environ["PP_LIB"]); _build.is_stale = lambda: False
from pytorch_points_amd import synthetic as S
from pytorch_points_amd._ext import sampling
from pytorch_points_amd.network.operations import ball_query
dev = torch.device("cuda:0")
B, N, C, ns = 32, 16384, 128, 64
x = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); c = x[:, ::4].contiguous()
idx = ball_query(0.1, ns, x, c)
go = torch.randn(B, C, c.shape[1], ns, device=dev)
out = sampling.group_points_grad(go, idx, N)
for _ in range(3): sampling.group_points_grad(go, idx, N)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): sampling.group_points_grad(go, idx, N)
b.record(); torch.cuda.synchronize()
print("group_points_grad %.4f ms  checksum %.6f" % (a.elapsed_time(b) / 10, float(out.double().sum())))
