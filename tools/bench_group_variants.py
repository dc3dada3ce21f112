#!/usr/bin/env python3
"""Interleaved timing of group_points kernel variants at config 4 in one process."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import sampling
variants = [int(v) for v in sys.argv[1:]] or [8, 108, 116, 132]
B, N, C, ns = 32, 16384, 128, 64
dev = torch.device("cuda:0")
x = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev)
centres = x[:, ::4].contiguous()
feats = torch.from_numpy(S.normal(2, (B, C, N))).to(dev)
idx = sampling.ball_query(centres, x, 0.1, ns)
setv = _lib.lib().pp_debug_set_group_points_variant
setv.argtypes = [ctypes.c_int]; setv.restype = None
times = {v: [] for v in variants}
ref = None
for rnd in range(5):
    for v in variants:
        setv(v)
        out = sampling.group_points(feats, idx)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            out = sampling.group_points(feats, idx)
        e1.record(); torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / 3)
        if ref is None: ref = out.clone()
        assert torch.equal(ref, out)
        del out
setv(0)
byt = 4.0 * B * C * N + 4.0 * B * idx.shape[1] * ns + 4.0 * B * C * idx.shape[1] * ns
for v in variants:
    t = np.median(times[v][1:])
    print("variant %4d  %.3f ms  %.2f TB/s  %.1f%% of 8 TB/s" % (v, t, byt / t / 1e9, byt / t / 1e9 / 80))
