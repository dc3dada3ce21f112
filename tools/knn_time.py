"""knn_points K = 1, 8, 16 and three_nn on the uniform sphere at the shapes of bench.py's other_ops_ms (PP_LIB: a variant library)"""
import os, sys, torch
sys.path.insert(0, ".")
if os.environ.get("PP_LIB"):
    from pytorch_points_amd import _build
    _build.LIB = os.path.abspath(os.environ["PP_LIB"]); _build.is_stale = lambda: False
from pytorch_points_amd import synthetic as S
from pytorch_points_amd._ext import sampling
from pytorch_points_amd.ops import knn_points
dev = torch.device("cuda:0")
B, N, M = 32, 16384, 4096
def run(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
x1 = torch.from_numpy(S.unit_sphere(2, B, N)).to(dev); x2 = torch.from_numpy(S.unit_sphere(3, B, N)).to(dev)
for rep in range(2):
    print("knn K=1 %.4f  K=8 %.4f  K=16 %.4f ms" % tuple(run(lambda: knn_points(x1, x2, K=K)) for K in (1, 8, 16)))
unknown = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); known = torch.from_numpy(S.unit_sphere(1, B, M)).to(dev)
d2 = torch.empty(B, N, 3, device=dev); idx = torch.empty(B, N, 3, dtype=torch.int32, device=dev)
print("three_nn %.4f ms" % run(lambda: sampling.three_nn_wrapper(B, N, M, unknown, known, d2, idx)))
