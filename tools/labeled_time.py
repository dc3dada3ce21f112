"""labeled Chamfer forward (grid search, one kernel) on several clouds: python tools/labeled_time.py  (PP_LIB: a variant library)"""
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
if os.environ.get("PP_LIB"):
    from pytorch_points_amd import _build
    _build.LIB = os.path.abspath(os.environ["PP_LIB"]); _build.is_stale = lambda: False
from pytorch_points_amd import synthetic as S
from pytorch_points_amd._ext import losses
import bench
dev = torch.device("cuda:0")
B, N = 32, 16384
def run(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for kind in ("sphere", "gaussian", "shapenet_like", "two_scales"):
    if kind == "sphere":
        x1, x2 = S.unit_sphere(0, B, N), S.unit_sphere(1, B, N)
    else:
        x1, x2 = bench._distribution(kind, 0, B, N), bench._distribution(kind, 1, B, N)
    rng = np.random.default_rng(3)
    for nl in (4, 16):
        l1 = torch.from_numpy(rng.integers(0, nl, (B, N)).astype(np.float32)).to(dev)
        l2 = torch.from_numpy(rng.integers(0, nl, (B, N)).astype(np.float32)).to(dev)
        t1, t2 = torch.from_numpy(np.ascontiguousarray(x1)).to(dev), torch.from_numpy(np.ascontiguousarray(x2)).to(dev)
        o = (torch.empty(B, N, device=dev), torch.empty(B, N, device=dev), torch.empty(B, N, dtype=torch.int32, device=dev),
             torch.empty(B, N, dtype=torch.int32, device=dev))
        print("%-14s %2d labels: labeled forward %.4f ms" % (kind, nl, run(lambda: losses.labeled_nmdistance_forward(t1, t2, l1, l2, *o))))
