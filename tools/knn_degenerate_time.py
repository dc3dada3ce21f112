"""knn_points / three_nn when ONE batch element has no usable grid (a NaN coordinate): the grid
kernel serves that element itself, a lane per query over all points.  Times the clean and the one-NaN case."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from pytorch_points_amd import synthetic as S
from pytorch_points_amd.ops import knn_points
from pytorch_points_amd.network.pointnet2_utils import three_nn
dev = torch.device("cuda:0")
B, N = 32, 16384
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
x1 = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); x2 = torch.from_numpy(S.unit_sphere(1, B, N)).to(dev)
x2n = x2.clone(); x2n[3, 777, 1] = float("nan")
kn = torch.from_numpy(S.unit_sphere(2, B, 4096)).to(dev); knn_ = kn.clone(); knn_[3, 77, 1] = float("nan")
for K in (8, 32):
    print("knn_points K=%d: clean %.3f ms, one NaN point in one cloud %.3f ms" % (K, t(lambda: knn_points(x1, x2, K=K)), t(lambda: knn_points(x1, x2n, K=K))))
print("three_nn: clean %.3f ms, one NaN point in one cloud %.3f ms" % (t(lambda: three_nn(x1, kn)), t(lambda: three_nn(x1, knn_))))
# the same degenerate element by itself: the grid kernel's in-kernel fallback against the scan kernels (debug knob: scan)
import ctypes
from pytorch_points_amd import _lib
L = _lib.lib()
for name in ("pp_debug_set_knn_search", "pp_debug_set_three_nn_search"):
    getattr(L, name).argtypes = [ctypes.c_int]; getattr(L, name).restype = None
y1, y2n, k1n = x1[3:4].contiguous(), x2n[3:4].contiguous(), knn_[3:4].contiguous()
for K in (8, 32):
    a = t(lambda: knn_points(y1, y2n, K=K))
    L.pp_debug_set_knn_search(1)
    b_ = t(lambda: knn_points(y1, y2n, K=K))
    L.pp_debug_set_knn_search(0)
    print("one element with a NaN point, knn K=%d: in the grid kernel %.3f ms, scan kernel %.3f ms" % (K, a, b_))
a = t(lambda: three_nn(y1, k1n))
L.pp_debug_set_three_nn_search(1)
b_ = t(lambda: three_nn(y1, k1n))
L.pp_debug_set_three_nn_search(0)
print("one element with a NaN point, three_nn: in the grid kernel %.3f ms, scan kernel %.3f ms" % (a, b_))
