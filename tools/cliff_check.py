"""grid search == brute force on clustered distributions at full cloud size (debug aid)"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import bench
from pytorch_points_amd import _lib
from pytorch_points_amd._ext import losses
dev = torch.device("cuda:0")
L = _lib.lib()
mode = L.pp_debug_set_nmdistance_search; mode.argtypes = [ctypes.c_int]; mode.restype = None
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
for kind in ("gaussian", "blobs8", "two_scales", "shapenet_like"):
    x1 = torch.from_numpy(bench._distribution(kind, 0, B, N)).to(dev); x2 = torch.from_numpy(bench._distribution(kind, 1, B, N)).to(dev)
    outs = []
    for m in (2, 2, 1):
        mode(m)
        d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
        i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
        losses.nmdistance_forward(x1, x2, d1, d2, i1, i2); torch.cuda.synchronize()
        outs.append((d1, d2, i1, i2))
    mode(0)
    same_runs = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    bad = [int((a != b).sum()) for a, b in zip(outs[0], outs[2])]
    print(kind, "two grid runs identical:", same_runs, "| mismatches vs brute force (d1,d2,i1,i2):", bad, flush=True)
