"""forward time of nndistance at config-2 size on several point distributions, for the forms of the search kernel
(pp_debug_set_nmdistance_tile: -1 the wave-private form of round 2, 256 / 512 / 768 queries per tile); every form is
compared bit for bit with the every-pair kernel.  python tools/tile_modes.py [kinds...]"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
import os
if os.environ.get("PP_LIB"):   # a variant of the library (tools/build_variant_lib.sh)
    from pytorch_points_amd import _build
    _build.LIB = os.path.abspath(os.environ["PP_LIB"])
    _build.is_stale = lambda: False
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import losses
dev = torch.device("cuda:0")
B, N = 32, 16384
rng = np.random.default_rng(0)
L = _lib.lib()
sett = L.pp_debug_set_nmdistance_tile; sett.argtypes = [ctypes.c_int]; sett.restype = None
sets = L.pp_debug_set_nmdistance_search; sets.argtypes = [ctypes.c_int]; sets.restype = None
tk = L.pp_debug_set_nmdistance_kernel_timing; tk.argtypes = [ctypes.c_int]; tk.restype = None
rd = L.pp_debug_nmdistance_kernel_ms3; rd.argtypes = [ctypes.POINTER(ctypes.c_float)] * 3; rd.restype = ctypes.c_int
def run(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def clouds(kind, seed):
    if kind == "sphere": return S.unit_sphere(seed, B, N)
    if kind == "same": return S.unit_sphere(0, B, N)   # both clouds identical: stage A settles every query
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
    if kind == "shapenet_like":
        x = rng.random((B, N, 3), dtype=np.float32) - 0.5
        q = N // 4
        x[:, :q, 2] = -0.5; x[:, q:2 * q, 0] = 0.2
        th = rng.random((B, q)) * 6.283; x[:, 2 * q:3 * q, 0] = 0.3 * np.cos(th); x[:, 2 * q:3 * q, 1] = 0.3 * np.sin(th)
        return x.astype(np.float32)
    if kind == "disjoint":
        x = rng.random((B, N, 3), dtype=np.float32)
        if seed: x += 5.0
        return x
kinds = sys.argv[1:] or ["sphere", "cube", "gaussian", "blobs8", "two_scales", "plane", "line", "shapenet_like", "disjoint"]
modes = tuple(int(v) for v in os.environ.get("PP_TILE_MODES", "-1,512,513,1024").split(","))
for kind in kinds:
    x1 = torch.from_numpy(np.ascontiguousarray(clouds(kind, 0))).to(dev); x2 = torch.from_numpy(np.ascontiguousarray(clouds(kind, 1))).to(dev)
    def outs():
        return (torch.empty(B, N, device=dev), torch.empty(B, N, device=dev),
                torch.empty(B, N, dtype=torch.int32, device=dev), torch.empty(B, N, dtype=torch.int32, device=dev))
    sets(1)
    ref = outs()
    losses.nmdistance_forward(x1, x2, *ref)
    sets(0)
    line = "%-14s" % kind
    ok = True
    for mode in modes:
        sett(mode)
        o = outs()
        ms = run(lambda: losses.nmdistance_forward(x1, x2, *o))
        tk(1)
        bms, sms, rms = [], [], []
        for _ in range(6):
            losses.nmdistance_forward(x1, x2, *o)
            a_, b_, c_ = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_float(0)
            rd(ctypes.byref(a_), ctypes.byref(b_), ctypes.byref(c_)); bms.append(a_.value); sms.append(b_.value); rms.append(c_.value)
        tk(0)
        same = all(torch.equal(a, b) for a, b in zip(o, ref))
        pend = ""
        if mode != -1:
            tot = (ctypes.c_uint * (2 * B))()
            wsb = _lib.cached_workspaces("nmdistance")[0]
            fn = L.pp_debug_nmdistance_pending; fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]; fn.restype = ctypes.c_int
            if fn(wsb.data_ptr(), B, N, N, tot) == 0:
                pend = " left %.2f%%" % (100.0 * sum(tot) / (2.0 * B * N))
        ok = ok and same
        line += " | %4d: fwd %.4f build %.4f stageA %.4f rest %.4f %s" % (mode, ms, np.mean(bms[1:]), np.mean(sms[1:]), np.mean(rms[1:]), ("ok" if same else "MISMATCH") + pend)
    sett(0)
    print(line, flush=True)
