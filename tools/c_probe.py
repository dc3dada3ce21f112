"""Chamfer forward at config-2 size for other point dimensions: time and fraction of the fp32 VALU roof
(lane-ops = pairs * (2 C + 1/2 + bookkeeping): C sub, C fma (the first is a multiply-add onto 0), 1/2 min3)"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import losses
dev = torch.device("cuda:0")
B, N = 32, 16384
setter = _lib.lib().pp_debug_set_nmdistance_variant; setter.argtypes = [ctypes.c_int]; setter.restype = None
for C in (2, 6, 9, 12, 16):
    x1 = torch.from_numpy(S.unit_sphere(0, B, N, C)).to(dev); x2 = torch.from_numpy(S.unit_sphere(1, B, N, C)).to(dev)
    d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
    i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
    res = []
    for variant in (0, 9):
        setter(variant)
        for _ in range(2): losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 5)
    setter(0)
    laneops = 2.0 * B * N * N * (2 * C + 0.5)
    print("C=%2d  tiled %.3f ms (VALU frac %.2f)  one-lane-per-query %.3f ms" % (C, res[0], laneops / (res[0] * 1e-3) / 78.6e12, res[1]))
