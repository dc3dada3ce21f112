"""forward time of the search operators on point distributions other than the uniform sphere"""
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
if os.environ.get("PP_LIB"):   # a variant of the library (tools/build_variant_lib.sh)
    from pytorch_points_amd import _build
    _build.LIB = os.path.abspath(os.environ["PP_LIB"])
    _build.is_stale = lambda: False
from pytorch_points_amd import synthetic as S
from pytorch_points_amd._ext import losses, sampling
from pytorch_points_amd.ops import knn_points
dev = torch.device("cuda:0")
B, N = 32, 16384
rng = np.random.default_rng(0)
def run(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5
def clouds(kind, seed):
    if kind == "sphere": return S.unit_sphere(seed, B, N)
    if kind == "cube": return rng.random((B, N, 3), dtype=np.float32)
    if kind == "gaussian": return rng.standard_normal((B, N, 3)).astype(np.float32)
    if kind == "blobs8": 
        c = rng.random((B, 8, 3), dtype=np.float32) * 2
        return (c[:, rng.integers(0, 8, N)] + rng.standard_normal((B, N, 3)).astype(np.float32) * 0.02).astype(np.float32)
    if kind == "two_scales":
        x = rng.random((B, N, 3), dtype=np.float32); x[:, : N // 2] *= 1e-2; return x
    if kind == "plane": 
        x = rng.random((B, N, 3), dtype=np.float32); x[..., 2] = 0.3; return x
    if kind == "line":
        x = np.zeros((B, N, 3), np.float32); x[..., 0] = rng.random((B, N), dtype=np.float32); return x
    if kind == "shapenet_like":   # thin surfaces: union of a few planes and a cylinder
        x = rng.random((B, N, 3), dtype=np.float32) - 0.5
        q = N // 4
        x[:, :q, 2] = -0.5; x[:, q:2 * q, 0] = 0.2
        th = rng.random((B, q)) * 6.283; x[:, 2 * q:3 * q, 0] = 0.3 * np.cos(th); x[:, 2 * q:3 * q, 1] = 0.3 * np.sin(th)
        return x.astype(np.float32)
d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
for kind in ("sphere", "cube", "gaussian", "blobs8", "two_scales", "plane", "line", "shapenet_like"):
    x1 = torch.from_numpy(np.ascontiguousarray(clouds(kind, 0))).to(dev); x2 = torch.from_numpy(np.ascontiguousarray(clouds(kind, 1))).to(dev)
    t_ch = run(lambda: losses.nmdistance_forward(x1, x2, d1, d2, i1, i2))
    c = x1[:, ::4].contiguous()
    ext = float((x1.amax(1) - x1.amin(1)).max())
    t_bq = run(lambda: sampling.ball_query(c, x1, 0.05 * ext, 32))
    t_kn = run(lambda: knn_points(c, x1, K=8))
    dd = torch.empty(B, N, 3, device=dev); ii = torch.empty(B, N, 3, dtype=torch.int32, device=dev)
    t_tn = run(lambda: sampling.three_nn_wrapper(B, N, 4096, x1, c, dd, ii))
    print("%-14s chamfer fwd %7.3f ms | ball_query %7.3f ms | knn8 %7.3f ms | three_nn %7.3f ms" % (kind, t_ch, t_bq, t_kn, t_tn))
