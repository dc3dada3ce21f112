"""nndistance forward (config 2's size) on the clouds of bench.py's other_distributions_fwd_ms, the group search's row
bitmap on (default) and off: ms per call, two passes each.  Usage: [PP_LIB=tools/libpp_hip_<tag>.so] python tools/far_time.py [kind ...]"""
import ctypes
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kinds = sys.argv[1:] or ["sphere", "cube", "gaussian", "blobs8", "two_scales", "shapenet_like", "disjoint", "shells", "lattice"]
sys.argv = sys.argv[:1]
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
if os.environ.get("PP_LIB"):  # a variant build of the library (tools/build_variant_lib.sh)
    from pytorch_points_amd import _build
    _build.LIB = os.path.abspath(os.environ["PP_LIB"])
    _build.is_stale = lambda: False
from pytorch_points_amd import _lib, synthetic as S  # noqa: E402
from pytorch_points_amd._ext import losses  # noqa: E402

B, N = 32, 16384
dev = torch.device("cuda:0")
knob = _lib.lib().pp_debug_set_nmdistance_row_bitmap
knob.argtypes = [ctypes.c_int]
knob.restype = None
d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)


def ms(a, b, n=20):
    for _ in range(5):
        losses.nmdistance_forward(a, b, d1, d2, i1, i2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        losses.nmdistance_forward(a, b, d1, d2, i1, i2)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for kind in kinds:
    if kind == "sphere":
        a = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); b = torch.from_numpy(S.unit_sphere(1, B, N)).to(dev)
    else:
        a = torch.from_numpy(bench._distribution(kind, 0, B, N)).to(dev); b = torch.from_numpy(bench._distribution(kind, 1, B, N)).to(dev)
    out = []
    for rep in range(2):
        for off in (0, 1):
            knob(off)
            out.append(ms(a, b))
    knob(0)
    print("%-14s row bitmap on %.4f %.4f ms   off %.4f %.4f ms" % (kind, out[0], out[2], out[1], out[3]), flush=True)
