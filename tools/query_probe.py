"""Phase clocks of grid_query_lds_kernel (a -DPP_QUERY_PROBE build of the library, made by this script into
/tmp; the shipped library carries no stamps).  python tools/query_probe.py [mode ...]"""
import ctypes, os, subprocess, sys, numpy as np, torch
sys.path.insert(0, ".")
from pytorch_points_amd import _build
lib = "/tmp/libpp_hip_probe.so"
cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *_build.HIPCC_FLAGS, "-DPP_QUERY_PROBE", "-I" + _build.INCLUDE, "-I" + _build.CSRC,
       *_build.sources(), "-o", lib]
subprocess.run(cmd, check=True)
_build.LIB = lib
_build.is_stale = lambda: False
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import losses
L = _lib.lib()
dev = torch.device("cuda:0")
B, N = 32, 16384
setq = L.pp_debug_set_nmdistance_stage_cap; setq.argtypes = [ctypes.c_int]; setq.restype = None
rd = L.pp_debug_read_query_phases; rd.argtypes = [ctypes.c_void_p]; rd.restype = ctypes.c_int
x1 = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); x2 = torch.from_numpy(S.unit_sphere(1, B, N)).to(dev)
d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
for mode in [int(a) for a in sys.argv[1:]] or [384, 320, 512]:
    setq(mode)
    for _ in range(5):
        losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
    b.record(); torch.cuda.synchronize()
    ph = np.zeros((8, 16), np.uint64)
    assert rd(ph.ctypes.data) == 0
    print("mode %d: fwd %.1f us" % (mode, a.elapsed_time(b) / 20 * 1e3))
    t0 = ph[:, 0].min()
    for w in range(8):
        d = (ph[w, 1:8].astype(np.int64) - ph[w, 0:7].astype(np.int64)) / 100.0
        print("   wg %4d start +%.2f us: " % (w * 512, (int(ph[w, 0]) - int(t0)) / 100.0) + " ".join("p%d %.2f" % (k, v) for k, v in enumerate(d)) +
              "  total %.2f" % ((int(ph[w, 7]) - int(ph[w, 0])) / 100.0))
