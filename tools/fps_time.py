"""Times furthest_point_sample through its three kernels (bucketed / CU cluster / one workgroup over all points) at
BASELINE config 3 and on clouds where the bucketed kernel's pruning is weakest.  Usage: python tools/fps_time.py"""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from pytorch_points_amd import _lib, synthetic as S  # noqa: E402
from pytorch_points_amd.network.geo_operations import furthest_point_sample  # noqa: E402

FORMS = {"default": 0, "single_block": 1, "cluster": 2, "bucket": 3}


def form(name):
    f = _lib.lib().pp_debug_set_fps_v1
    f.argtypes = [ctypes.c_int]
    f.restype = None
    f(FORMS[name])


def chain(one_pick):
    """the bucketed kernel's chain: 2 = several independent picks per round, 1 = one pick per round"""
    f = _lib.lib().pp_debug_set_fps_bucket_chain
    f.argtypes = [ctypes.c_int]
    f.restype = None
    f(1 if one_pick else 2)


def clouds(B, N):
    rng = np.random.default_rng(5)
    out = {"sphere": S.unit_sphere(0, B, N)}
    out["gaussian"] = S.normal(1, (B, N, 3))
    out["cube"] = (S.uniform01(2, (B, N, 3)).reshape(B, N, 3)).astype(np.float32)
    c = rng.normal(size=(B, 8, 3)).astype(np.float32) * 3
    sel = rng.integers(0, 8, (B, N))
    out["blobs8"] = (np.take_along_axis(c, sel[..., None].repeat(3, -1), 1) + 0.02 * S.normal(3, (B, N, 3))).astype(np.float32)
    two = S.unit_sphere(4, B, N).copy()
    two[:, : N // 2] = two[:, : N // 2] * 0.01 + 0.3
    out["two_scales"] = two
    pl = S.normal(6, (B, N, 3)); pl[..., 2] = 0
    out["plane"] = pl
    return out


def time_it(x, m, reps=3):
    furthest_point_sample(x, m, NCHW=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        furthest_point_sample(x, m, NCHW=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"   # the bucketed kernel's two chains at config 3 only
    shapes = [(16, 65536, 4096), (16, 16384, 1024), (32, 8192, 512), (64, 4096, 1024), (1, 65536, 4096), (4, 262144, 4096)]
    for (B, N, m) in shapes[:1] if quick else shapes:
        cl = clouds(B, N)
        for name, x in cl.items():
            if name != "sphere" and (B, N) != (16, 65536):
                continue
            xt = torch.from_numpy(np.ascontiguousarray(x.astype(np.float32))).to(dev)
            row = {}
            ref = None
            for f in ("bucket", "bucket_one_pick") if quick else ("bucket", "bucket_one_pick", "cluster", "single_block"):
                if f == "single_block" and (N * B > 16 * 65536 or (name != "sphere")):
                    continue
                form("bucket" if f == "bucket_one_pick" else f)
                chain(f == "bucket_one_pick")
                try:
                    row[f] = time_it(xt, m, reps=5 if f.startswith("bucket") else 2)
                    idx = furthest_point_sample(xt, m, NCHW=False)[0]
                    if ref is None:
                        ref = idx
                    else:
                        assert torch.equal(ref, idx), (name, f)
                finally:
                    form("default")
                    _lib.lib().pp_debug_set_fps_bucket_chain(0)
            print("B=%d N=%d m=%d %-10s " % (B, N, m, name) +
                  "  ".join("%s %.3f ms (%.3f us/pick)" % (k, v, v * 1e3 / (m - 1)) for k, v in row.items()), flush=True)


if __name__ == "__main__":
    main()
