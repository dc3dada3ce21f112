"""python tools/fwd_loop.py TILE_MODE (pp_debug_set_nmdistance_tile) [kind] [iters]: nndistance forward at config 2 in a loop (for rocprofv3)"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import losses
dev = torch.device("cuda:0")
B, N = 32, 16384
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
L = _lib.lib()
setq = L.pp_debug_set_nmdistance_tile; setq.argtypes = [ctypes.c_int]; setq.restype = None
setq(mode)
import os
if os.environ.get('PP_BUILD_GENERAL') == '1':   # the general build instead of the LDS-sorted one (A/B of their counters)
    sb = L.pp_debug_set_nmdistance_build; sb.argtypes = [ctypes.c_int]; sb.restype = None; sb(1)
x1 = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); x2 = torch.from_numpy(S.unit_sphere(1, B, N)).to(dev)
d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
gx1 = torch.empty_like(x1); gx2 = torch.empty_like(x2)
g1 = torch.full((B, N), 1.0 / (B * N), device=dev)
for _ in range(iters):
    losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
    losses.nmdistance_backward(x1, x2, gx1, gx2, g1, g1, i1, i2)
torch.cuda.synchronize()
