"""per-call forward times on the shell-vs-core family: every pair, and the default search call after call (the first call
finds the direction unprunable in the stage-A launch's tail; later ones route it in front of the build)"""
import ctypes, importlib.util, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from pytorch_points_amd import _lib
from pytorch_points_amd._ext import losses
dev = torch.device("cuda:0"); B, N = 32, 16384
kind = sys.argv[1] if len(sys.argv) > 1 else "shell_vs_core"
a = torch.from_numpy(bench._distribution(kind, 0, B, N)).to(dev); b = torch.from_numpy(bench._distribution(kind, 1, B, N)).to(dev)
d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
mode = _lib.lib().pp_debug_set_nmdistance_search; mode.argtypes = [ctypes.c_int]; mode.restype = None
def one():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); losses.nmdistance_forward(a, b, d1, d2, i1, i2); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
mode(1); print("every pair:", ["%.3f" % one() for _ in range(6)]); mode(0)
print("default    :", ["%.3f" % one() for _ in range(10)])
# one direction at a time: swap in an ordinary cloud for the other
