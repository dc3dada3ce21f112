"""nndistance forward on the adversarial cloud families of bench.py's other_distributions_fwd_ms: time (default search and
the every-pair kernel) and bit-equality of the two.  Usage: python tools/adversarial_time.py [B N]"""
import ctypes
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
sys.argv = sys.argv[:1] + sys.argv[1:]
spec.loader.exec_module(bench)
from pytorch_points_amd import _lib  # noqa: E402
from pytorch_points_amd._ext import losses  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    dev = torch.device("cuda:0")
    mode = _lib.lib().pp_debug_set_nmdistance_search
    mode.argtypes = [ctypes.c_int]
    mode.restype = None
    d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
    i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
    for kind in ("cube", "gaussian", "blobs8", "two_scales", "shapenet_like", "disjoint", "shells", "shell_vs_core",
                 "identical", "lattice"):
        a = torch.from_numpy(bench._distribution(kind, 0, B, N)).to(dev)
        b = torch.from_numpy(bench._distribution(kind, 1, B, N)).to(dev)
        res = {}
        outs = {}
        for name, m in (("default", 0), ("every_pair", 1)):
            mode(m)
            try:
                for _ in range(2):
                    losses.nmdistance_forward(a, b, d1, d2, i1, i2)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    losses.nmdistance_forward(a, b, d1, d2, i1, i2)
                e1.record()
                torch.cuda.synchronize()
                res[name] = e0.elapsed_time(e1) / 5
                outs[name] = (d1.clone(), d2.clone(), i1.clone(), i2.clone())
            finally:
                mode(0)
        same = all(torch.equal(x, y) for x, y in zip(outs["default"], outs["every_pair"]))
        print("%-14s default %.3f ms   every pair %.3f ms   outputs %s" % (kind, res["default"], res["every_pair"],
                                                                          "bit-identical" if same else "DIFFER"), flush=True)


if __name__ == "__main__":
    main()
