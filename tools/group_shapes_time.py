#!/usr/bin/env python3
"""group_points over shape classes: the library's choice against forced forms (which forms earn their keep).
usage: python tools/group_shapes_time.py [variant ...]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import sampling
variants = [int(v) for v in sys.argv[1:]] or [0, 616, 608, 604, 8, 4, 2, 1]
dev = torch.device("cuda:0")
setv = _lib.lib().pp_debug_set_group_points_variant
setv.argtypes = [ctypes.c_int]; setv.restype = None
shapes = [(32, 128, 16384, 4096, 64), (32, 16, 16384, 4096, 64), (16, 32, 8192, 2048, 32), (8, 64, 8192, 1024, 32),
          (32, 6, 16384, 4096, 64), (16, 64, 4096, 1024, 64), (4, 128, 65536, 4096, 32), (32, 3, 16384, 4096, 64),
          (8, 32, 10001, 2048, 33), (16, 128, 1024, 512, 32), (32, 128, 16383, 4096, 63), (16, 64, 8192, 1024, 16),
          (16, 64, 8192, 512, 16), (8, 64, 20000, 3000, 32)]
for (B, C, N, npoint, ns) in shapes:
    feats = torch.from_numpy(S.normal(2, (B, C, N))).to(dev)
    idx = torch.from_numpy((S.uniform01(3, (B, npoint, ns)).reshape(B, npoint, ns) * N).astype(np.int32)).to(dev)
    row = []
    ref = None
    out = torch.empty(B, C, npoint, ns, device=dev)      # (one output buffer: the allocator stays out of the timing)
    L = _lib.lib()
    for v in variants:
        setv(v)
        try:
            with _lib.on_device(dev) as stream:
                call = lambda: _lib.check(L.pp_group_points_f32(_lib.ptr(feats), _lib.ptr(idx), _lib.ptr(out), B, C, N, npoint, ns, stream), "group_points")
                call()
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    call()
                e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 5
            if ref is None:
                ref = out.clone()
            else:
                assert torch.equal(ref, out)
            row.append("%d: %.3f" % (v, t))
        finally:
            setv(0)
    byt = 4.0 * B * C * npoint * ns
    print("B=%d C=%d N=%d npoint=%d ns=%d (%.0f MB out)  ms  %s" % (B, C, N, npoint, ns, byt / 1e6, "  ".join(row)), flush=True)
