"""forward of nndistance at config-2 size with an odd point count (every batch element's cloud starts at another
4-byte phase: the unaligned build kernel) beside the aligned one: build / stage A / rest kernel times"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import losses
dev = torch.device("cuda:0")
L = _lib.lib()
tk = L.pp_debug_set_nmdistance_kernel_timing; tk.argtypes = [ctypes.c_int]; tk.restype = None
rd = L.pp_debug_nmdistance_kernel_ms3; rd.argtypes = [ctypes.POINTER(ctypes.c_float)] * 3; rd.restype = ctypes.c_int
for n in (16384, 16383, 16381):
    B = 32
    x1 = torch.from_numpy(S.unit_sphere(0, B, n)).to(dev); x2 = torch.from_numpy(S.unit_sphere(1, B, n)).to(dev)
    o = (torch.empty(B, n, device=dev), torch.empty(B, n, device=dev), torch.empty(B, n, dtype=torch.int32, device=dev), torch.empty(B, n, dtype=torch.int32, device=dev))
    for _ in range(3): losses.nmdistance_forward(x1, x2, *o)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): losses.nmdistance_forward(x1, x2, *o)
    b.record(); torch.cuda.synchronize()
    tk(1); acc = []
    for _ in range(6):
        losses.nmdistance_forward(x1, x2, *o)
        v = [ctypes.c_float(0) for _ in range(3)]; rd(*[ctypes.byref(q) for q in v]); acc.append([q.value for q in v])
    tk(0)
    m = np.mean(acc[1:], 0)
    print("N=M=%d: fwd %.4f ms  build %.4f stageA %.4f rest %.4f" % (n, a.elapsed_time(b) / 20, m[0], m[1], m[2]), flush=True)
