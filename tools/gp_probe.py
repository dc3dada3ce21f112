"""group_points forward timing per kernel variant (0 = automatic, 116 = DMA V=16, 132 = DMA V=32 packed idx)"""
import ctypes, sys, torch
sys.path.insert(0, ".")
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import sampling
dev = torch.device("cuda:0")
B, N, M, ns, C = 32, 16384, 4096, 64, 128
x = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev)
c = x[:, ::4].contiguous()
idx = sampling.ball_query(c, x, 0.1, ns)
f = torch.randn(B, C, N, device=dev)
var = _lib.lib().pp_debug_set_group_points_variant
var.argtypes = [ctypes.c_int]; var.restype = None
ref = None
for v in [int(a) for a in sys.argv[1:]] or [0, 116, 132]:
    var(v)
    for _ in range(2): out = sampling.group_points(f, idx)
    torch.cuda.synchronize()
    if ref is None: ref = out.clone()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = sampling.group_points(f, idx)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("variant %d: %.3f ms (%.2f TB/s of output) equal=%s" % (v, ms, 4.295 / ms, torch.equal(out, ref)))
var(0)
