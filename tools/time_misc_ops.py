#!/usr/bin/env python3
"""Timings of the ops that are not BASELINE configs (so that nothing on the path is left
embarrassingly slow): three_nn, three_interpolate fwd/bwd, gather fwd/bwd, labeled Chamfer."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("PP_LIB"):   # a variant of the library (tools/build_variant_lib.sh)
    from pytorch_points_amd import _build
    _build.LIB = os.path.abspath(os.environ["PP_LIB"]); _build.is_stale = lambda: False
from pytorch_points_amd import synthetic as S
from pytorch_points_amd._ext import sampling, losses
dev = torch.device("cuda:0")
def t(fn, n=5):
    n = max(n, 4) * 4   # (5 calls untimed, 4 n timed: three calls after one were the clocks settling as much as the op)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
B, N, M, C = 32, 16384, 4096, 128
unknown = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev)
known = torch.from_numpy(S.unit_sphere(1, B, M)).to(dev)
d2 = torch.empty(B, N, 3, device=dev); idx = torch.empty(B, N, 3, dtype=torch.int32, device=dev)
ms = t(lambda: sampling.three_nn_wrapper(B, N, M, unknown, known, d2, idx))
print("three_nn B=%d N=%d M=%d: %.3f ms  (%.2f Tpairs/s)" % (B, N, M, ms, B * N * M / ms / 1e9))
feats = torch.randn(B, C, M, device=dev)
w = torch.rand(B, N, 3, device=dev); w /= w.sum(-1, keepdim=True)
out = torch.empty(B, C, N, device=dev)
ms = t(lambda: sampling.three_interpolate_wrapper(B, C, M, N, feats, idx, w, out))
byt = 4.0 * B * C * (M + N) + 24.0 * B * N
print("three_interpolate C=%d: %.3f ms  (%.2f TB/s algorithmic)" % (C, ms, byt / ms / 1e9))
gp = torch.zeros(B, C, M, device=dev)
import ctypes
from pytorch_points_amd import _lib
_v = _lib.lib().pp_debug_set_three_interpolate_grad_variant
_v.argtypes = [ctypes.c_int]; _v.restype = None
for name, v in (("auto", 0), ("lds columns f64", 2), ("sorted triples", 3)):
    _v(v)
    ms = t(lambda: sampling.three_interpolate_grad_wrapper(B, C, N, M, out, idx, w, gp))
    print("three_interpolate_grad [%s]: %.3f ms  (%.2f G adds/s)" % (name, ms, 3.0 * B * C * N / ms / 1e6))
_v(0)
gi = torch.randint(0, N, (B, M), dtype=torch.int32, device=dev)
f2 = torch.randn(B, C, N, device=dev); go = torch.empty(B, C, M, device=dev)
ms = t(lambda: sampling.gather_forward(B, C, N, M, f2, gi, go)); print("gather_forward C=%d N=%d->%d: %.3f ms" % (C, N, M, ms))
gg = torch.zeros(B, C, N, device=dev)
ms = t(lambda: sampling.gather_backward(B, C, N, M, go, gi, gg)); print("gather_backward: %.3f ms" % ms)
x1 = torch.from_numpy(S.unit_sphere(2, B, N)).to(dev); x2 = torch.from_numpy(S.unit_sphere(3, B, N)).to(dev)
torch.manual_seed(3)
l1 = torch.randint(0, 4, (B, N), device=dev).float(); l2 = torch.randint(0, 4, (B, N), device=dev).float()
dd1 = torch.empty(B, N, device=dev); dd2 = torch.empty(B, N, device=dev)
ii1 = torch.empty(B, N, dtype=torch.int32, device=dev); ii2 = torch.empty(B, N, dtype=torch.int32, device=dev)
ms = t(lambda: losses.labeled_nmdistance_forward(x1, x2, l1, l2, dd1, dd2, ii1, ii2), 3)
print("labeled_nmdistance_forward B=%d N=M=%d: %.3f ms  (%.2f Tpairs/s)" % (B, N, ms, 2.0 * B * N * N / ms / 1e9))
from pytorch_points_amd.ops import knn_points
_ks = _lib.lib().pp_debug_set_knn_search
_ks.argtypes = [ctypes.c_int]; _ks.restype = None
for K in (1, 8, 16):
    for name, v in (("grid", 0), ("scan", 1)):
        _ks(v)
        ms = t(lambda: knn_points(x1, x2, K=K), 3)
        print("knn_points K=%d B=%d N=M=%d [%s]: %.3f ms" % (K, B, N, name, ms))
_ks(0)
