"""Phase clocks of grid_build_kernel (a -DPP_BUILD_PROBE build of the library: PP_PROBE_FLAGS="-DPP_BUILD_PROBE" bash
tools/build_probe_lib.sh) on a distribution of bench.py: per phase, the mean over the launch's workgroups and the
workgroup that takes longest.  python tools/build_phases.py [kind ...]"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, ".")
from pytorch_points_amd import _build
_build.LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("PP_PROBE_LIB", "libpp_hip_probe.so"))
_build.is_stale = lambda: False
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import losses
import bench
L = _lib.lib()
dev = torch.device("cuda:0")
B, N = 32, 16384
rd = L.pp_debug_read_build_phases; rd.argtypes = [ctypes.c_void_p]; rd.restype = ctypes.c_int
for kind in sys.argv[1:] or ["sphere", "two_scales", "blobs8"]:
    if kind == "sphere":
        x1, x2 = S.unit_sphere(0, B, N), S.unit_sphere(1, B, N)
    elif kind == "cube":
        r = np.random.default_rng(0)
        x1, x2 = r.random((B, N, 3), dtype=np.float32), r.random((B, N, 3), dtype=np.float32)
    else:
        x1, x2 = bench._distribution(kind, 0, B, N), bench._distribution(kind, 1, B, N)
    x1, x2 = torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev)
    o = (torch.empty(B, N, device=dev), torch.empty(B, N, device=dev), torch.empty(B, N, dtype=torch.int32, device=dev), torch.empty(B, N, dtype=torch.int32, device=dev))
    for _ in range(4): losses.nmdistance_forward(x1, x2, *o)
    torch.cuda.synchronize()
    ph = np.zeros((512, 16), np.uint64)
    assert rd(ph.ctypes.data) == 0
    ph = ph[:256].astype(np.int64)
    d = np.diff(ph[:, :11], axis=1) / 100.0
    tot = (ph[:, 10] - ph[:, 0]) / 100.0
    w = int(tot.argmax())
    print("%-12s build: workgroup life mean %.1f max %.1f us (wg %d); launch %.1f us" % (kind, tot.mean(), tot.max(), w, (ph[:, 10].max() - ph[:, 0].min()) / 100.0))
    print("   mean per phase: " + " ".join("p%d %.1f" % (k, v) for k, v in enumerate(d.mean(0))))
    print("   slowest wg    : " + " ".join("p%d %.1f" % (k, v) for k, v in enumerate(d[w])))
