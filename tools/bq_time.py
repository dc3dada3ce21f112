"""ball_query at config 4 (B=32, N=16384, npoint=4096, r=0.1, nsample=64): time and equality with the scan kernel"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, ".")
if os.environ.get("PP_LIB"):
    from pytorch_points_amd import _build
    _build.LIB = os.path.abspath(os.environ["PP_LIB"]); _build.is_stale = lambda: False
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd.network.operations import ball_query
dev = torch.device("cuda:0")
B, N, ns, r = 32, 16384, 64, 0.1
x = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); c = x[:, ::4].contiguous()
knob = _lib.lib().pp_debug_set_ball_query_search; knob.argtypes = [ctypes.c_int]; knob.restype = None
knob(1); ref = ball_query(r, ns, x, c); knob(0)
lpc = _lib.lib().pp_debug_set_ball_query_lpc; lpc.argtypes = [ctypes.c_int]; lpc.restype = None
lpc(int(os.environ.get("PP_BQ_LPC", "0")))
got = ball_query(r, ns, x, c)
for _ in range(5): ball_query(r, ns, x, c)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): ball_query(r, ns, x, c)
b.record(); torch.cuda.synchronize()
print("ball_query %.4f ms  %s" % (a.elapsed_time(b) / 50, "ok" if torch.equal(ref, got) else "MISMATCH"))
