#!/usr/bin/env python3
"""Interleaved timing of the Chamfer forward kernel variants in ONE process (guide rule 24).
usage: python tools/sweep_chamfer.py [variants...]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import losses

variants = [int(v) for v in sys.argv[1:]] or [4, 8, 416, 1004, 1008, 1416, 1816]
B, N = 32, 16384
dev = torch.device("cuda:0")
x1 = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev)
x2 = torch.from_numpy(S.unit_sphere(1, B, N)).to(dev)
d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
setv = _lib.lib().pp_debug_set_nmdistance_variant
setv.argtypes = [ctypes.c_int]; setv.restype = None
ref = None
times = {v: [] for v in variants}
for rnd in range(6):
    for v in variants:
        setv(v)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
        e0.record()
        for _ in range(5):
            losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
        e1.record(); torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / 5)
        cur = (d1.clone(), i1.clone(), d2.clone(), i2.clone())
        if ref is None: ref = cur
        assert all(torch.equal(a, b) for a, b in zip(ref, cur)), "variant %d differs" % v
setv(0)
# the exact grid search (default operator path when a workspace is given)
L = _lib.lib()
nb = int(L.pp_nmdistance_forward_workspace_bytes(B, N, N, 3))
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
def grid():
    _lib.check(L.pp_nmdistance_forward_ws_f32(_lib.ptr(x1), _lib.ptr(x2), _lib.ptr(d1), _lib.ptr(i1), _lib.ptr(d2), _lib.ptr(i2),
                                              B, N, N, 3, _lib.ptr(ws), nb, None), "grid")
torch.cuda.synchronize()
with torch.cuda.stream(torch.cuda.default_stream()):
    grid(); torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(ref, (d1, i1, d2, i2))), "grid search differs from brute force"
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): grid()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
    print("grid search      median %.3f ms  %.2f Tpairs/s equivalent" % (np.median(ts), 2 * B * N * N / np.median(ts) / 1e9))
for v in variants:
    t = np.array(times[v][1:])
    print("variant %5d  median %.3f ms  min %.3f ms   %.2f Tpairs/s" % (v, np.median(t), t.min(), 2 * B * N * N / np.median(t) / 1e9))
