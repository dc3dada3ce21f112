"""group_points_grad timing for index patterns: ball rows with pads (r=0.1), full ball rows (r=0.25), random"""
import ctypes, sys, torch
sys.path.insert(0, ".")
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import sampling
dev = torch.device("cuda:0")
B, N, M, ns, C = 32, 16384, 4096, 64, 128
x = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev)
c = x[:, ::4].contiguous()
go = torch.randn(B, C, M, ns, device=dev)
var = _lib.lib().pp_debug_set_group_points_grad_variant
var.argtypes = [ctypes.c_int]; var.restype = None
smode = _lib.lib().pp_debug_set_scatter_mode
smode.argtypes = [ctypes.c_int]; smode.restype = None
smode(1)
pats = {"ball r=0.10 (pads)": sampling.ball_query(c, x, 0.1, ns), "ball r=0.25 (full)": sampling.ball_query(c, x, 0.25, ns),
        "random": torch.randint(0, N, (B, M, ns), device=dev, dtype=torch.int32)}
for v in [int(a) for a in sys.argv[1:]] or [2, 3]:
    var(v)
    for name, idx in pats.items():
        for _ in range(2): sampling.group_points_grad(go, idx, N)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): sampling.group_points_grad(go, idx, N)
        e1.record(); torch.cuda.synchronize()
        print("variant %d  %-20s %.3f ms" % (v, name, e0.elapsed_time(e1) / 5))
