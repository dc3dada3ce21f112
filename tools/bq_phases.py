"""Step clocks of bq_query_kernel (a -DPP_BQ_PROBE build: SRC=ball_grid bash tools/build_variant_lib.sh bqprobe -DPP_BQ_PROBE):
per wave the time of each step and the launch's timeline.  PP_LIB=tools/libpp_hip_bqprobe.so python tools/bq_phases.py"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, ".")
from pytorch_points_amd import _build
_build.LIB = os.path.abspath(os.environ.get("PP_LIB", "tools/libpp_hip_bqprobe.so")); _build.is_stale = lambda: False
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import sampling
dev = torch.device("cuda:0")
B, N, ns = 32, 16384, 64
x = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); c = x[:, ::4].contiguous()
for _ in range(4): idx = sampling.ball_query(c, x, 0.1, ns)
torch.cuda.synchronize()
L = _lib.lib(); rd = L.pp_debug_read_bq_phases; rd.argtypes = [ctypes.c_void_p]; rd.restype = ctypes.c_int
ph = np.zeros((16384, 8), np.uint64); assert rd(ph.ctypes.data) == 0
nb = B * (4096 // 16); ph = ph[:nb].astype(np.int64)
d = np.diff(ph[:, :7], axis=1) / 100.0
names = ["centre + cell marks", "point marks", "ranks + candidate ids", "gather", "scan", "rows out"]
print("per wave, us: " + " | ".join("%s %.2f" % (n, v) for n, v in zip(names, d.mean(0))) + " | life %.2f" % ((ph[:, 6] - ph[:, 0]).mean() / 100.0))
t0 = ph[:, 0].min(); st, en = (ph[:, 0] - t0) / 100.0, (ph[:, 6] - t0) / 100.0
print("launch: last start %.1f, last end %.1f us; waves alive at t: " % (st.max(), en.max()) + " ".join("%d:%d" % (t, int(((st <= t) & (en > t)).sum())) for t in range(0, int(en.max()) + 1, 8)))
