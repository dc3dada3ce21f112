"""config 3 (B=16, N=65536 -> 4096) and three other clouds through the bucketed FPS kernel of the library named by PP_LIB
(variants of the bucket -> wave dealing: SRC=fps_bucket tools/build_variant_lib.sh fskew<s> -DPP_FPSB_SKEW=<s>); picks
compared with the default library's are the caller's business (the test-suite does it for the shipped one)"""
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
if os.environ.get("PP_LIB"):
    from pytorch_points_amd import _build
    _build.LIB = os.path.abspath(os.environ["PP_LIB"]); _build.is_stale = lambda: False
from pytorch_points_amd import synthetic as S
from pytorch_points_amd.network.geo_operations import furthest_point_sample
import bench
dev = torch.device("cuda:0")
B, N, m = 16, 65536, 4096
out = []
for kind in ("sphere", "gaussian", "cube", "blobs8"):
    if kind == "sphere": x = S.unit_sphere(0, B, N)
    elif kind == "cube": x = np.random.default_rng(5).random((B, N, 3), dtype=np.float32)
    else: x = bench._distribution(kind, 0, B, N)
    x = torch.from_numpy(x).to(dev)
    for _ in range(2): idx, _pc = furthest_point_sample(x, m, NCHW=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): idx, _pc = furthest_point_sample(x, m, NCHW=False)
    b.record(); torch.cuda.synchronize()
    out.append("%s %.3f ms (checksum %d)" % (kind, a.elapsed_time(b) / 3, int(idx.long().sum())))
print(os.environ.get("PP_LIB", "default"), " | ".join(out))
