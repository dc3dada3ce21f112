"""Phase clocks of grid_query_wave_kernel (p0 rows, p1 region, p2 copy, p3 stage A, p4 second level, p5 lane cubes,
p6 whole-wave cubes, p7 group search, p8 the rest) (a -DPP_QUERY_PROBE build of the library,
tools/libpp_hip_probe.so; the shipped library carries no stamps).  python tools/query_probe.py [tile mode ...]
(modes: pp_debug_set_nmdistance_tile: -1 wave-private form, 256 / 512 / 768 queries per tile)"""
import ctypes, os, subprocess, sys, numpy as np, torch
sys.path.insert(0, ".")
from pytorch_points_amd import _build
# built in the container with tools/build_probe_lib.sh (hipcc takes minutes for the whole library: not on the GPU box)
lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("PP_PROBE_LIB", "libpp_hip_probe.so"))
if not os.path.exists(lib):
    subprocess.run(["bash", os.path.join(os.path.dirname(os.path.abspath(__file__)), "build_probe_lib.sh")], check=True)
_build.LIB = lib
_build.is_stale = lambda: False
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import losses
L = _lib.lib()
dev = torch.device("cuda:0")
B, N = 32, 16384
setq = L.pp_debug_set_nmdistance_tile; setq.argtypes = [ctypes.c_int]; setq.restype = None
rd = L.pp_debug_read_query_phases; rd.argtypes = [ctypes.c_void_p]; rd.restype = ctypes.c_int
import bench
kind = os.environ.get("PP_PROBE_KIND", "sphere")
if kind == "sphere":
    x1 = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); x2 = torch.from_numpy(S.unit_sphere(1, B, N)).to(dev)
else:
    x1 = torch.from_numpy(bench._distribution(kind, 0, B, N)).to(dev); x2 = torch.from_numpy(bench._distribution(kind, 1, B, N)).to(dev)
d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
for mode in [int(a) for a in sys.argv[1:]] or [0]:
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
    wv = np.zeros((1 << 17, 10), np.uint32)
    rdw = L.pp_debug_read_query_wave_phases; rdw.argtypes = [ctypes.c_void_p]; rdw.restype = ctypes.c_int
    assert rdw(wv.ctypes.data) == 0
    nw = min(1 << 17, ((B * ((N + 255) // 256) * 2 + 7) // 8) * 8 * 4)
    wv = wv[:nw, 1:].astype(np.float64) / 100.0
    print("   per wave, us:  " + " ".join("p%d mean %.1f max %.1f@wave%d |" % (k, wv[:, k].mean(), wv[:, k].max(), wv[:, k].argmax()) for k in range(9)))
    gs = np.zeros(8, np.uint64)
    rdg = L.pp_debug_read_query_group_stats; rdg.argtypes = [ctypes.c_void_p, ctypes.c_int]; rdg.restype = ctypes.c_int
    assert rdg(gs.ctypes.data, 1) == 0
    print("   group search (all launches so far): calls %d groups %d blind %d candidates %d (max per call %d) rows %d, rows listed %d, candidates walked %d" % tuple(int(x) for x in gs[:8]))
    if hasattr(L, "pp_debug_read_query_group_times"):
        gt = np.zeros(8, np.uint64)
        rgt = L.pp_debug_read_query_group_times; rgt.argtypes = [ctypes.c_void_p, ctypes.c_int]; rgt.restype = ctypes.c_int
        if rgt(gt.ctypes.data, 1) == 0 and gs[0] > 0:
            print("   group search, us per call: sampling %.1f grouping %.1f row list %.1f row cuts+spans %.1f fetch+sift %.1f walk %.1f" % tuple(float(x) / 100.0 / float(gs[0]) for x in gt[:6]))
    sp = np.zeros((1 << 17, 2), np.uint32)
    rds = L.pp_debug_read_query_wave_span; rds.argtypes = [ctypes.c_void_p]; rds.restype = ctypes.c_int
    if rds(sp.ctypes.data) == 0:
        sp = sp[:nw].astype(np.int64)
        ok = sp[:, 1] >= sp[:, 0]
        t0w = sp[ok, 0].min()
        st, en = (sp[:, 0] - t0w) / 100.0, (sp[:, 1] - t0w) / 100.0
        life = en - st
        print("   timeline (us after the first wave's start): last start %.1f, last end %.1f; waves alive at t: " % (st[ok].max(), en[ok].max()) +
              " ".join("%d:%d" % (tt, int(((st <= tt) & (en > tt) & ok).sum())) for tt in range(0, int(en[ok].max()) + 1, max(1, int(en[ok].max()) // 12))))
        order = np.argsort(-life)[:12]
        print("   longest waves: " + " ".join("w%d[%.0f-%.0f]" % (w, st[w], en[w]) for w in order))
        gw = None
        if hasattr(L, "pp_debug_read_query_wave_groups"):
            gw = np.zeros((1 << 17, 16), np.uint32)
            rgw = L.pp_debug_read_query_wave_groups; rgw.argtypes = [ctypes.c_void_p]; rgw.restype = ctypes.c_int
            assert rgw(gw.ctypes.data) == 0
        for w in order:
            print("      w%d phases (us): " % w + " ".join("p%d %.0f" % (k, wv[w, k]) for k in range(9)) +
                  ("" if gw is None else "   group search: open lanes %d groups %d blind %d candidates %d walked %d rows listed %d" % (
                      gw[w, 5], gw[w, 0], gw[w, 1], gw[w, 2], gw[w, 3], gw[w, 4]) +
                   " | first group: %d members, box %.3f x %.3f x %.3f, sqrt(us) %.3f sqrt(U) %.3f, h %.3f" % (
                      gw[w, 11], *[float(x) for x in gw[w, 6:11].view(np.float32)], float(gw[w, 12:13].view(np.float32)[0]))))
        if gw is not None:
            g = gw[:nw].astype(np.float64)
            p7 = wv[:, 7]
            ext = gw[:nw, 6:9].copy().view(np.float32).max(1).astype(np.float64)   # largest extent of the first group's box
            mem = g[:, 11]
            lifes = wv.sum(1)
            for lo, hi in ((0, 0.02), (0.02, 0.04), (0.04, 0.06), (0.06, 0.08), (0.08, 0.12), (0.12, 1e9)):
                sel = (ext >= lo) & (ext < hi) & (g[:, 0] > 0)
                if sel.any():
                    print("      first group's box extent in [%g, %g): %d waves; life mean %.1f p90 %.1f max %.1f us; members %.1f groups %.1f" % (
                        lo, hi, int(sel.sum()), lifes[sel].mean(), np.percentile(lifes[sel], 90), lifes[sel].max(), mem[sel].mean(), g[sel, 0].mean()))
            for lo, hi in ((1, 2), (2, 3), (3, 4), (4, 99)):
                sel = (g[:, 0] >= lo) & (g[:, 0] < hi)
                if sel.any():
                    print("      groups in [%d, %d): %d waves; life mean %.1f p90 %.1f max %.1f us" % (lo, hi, int(sel.sum()), lifes[sel].mean(), np.percentile(lifes[sel], 90), lifes[sel].max()))
            for lo, hi in ((0, 20), (20, 50), (50, 100), (100, 200), (200, 1e9)):
                sel = (p7 >= lo) & (p7 < hi)
                if sel.any():
                    print("      waves with p7 in [%g, %g) us: %d; mean open lanes %.1f groups %.1f blind %.1f candidates %.0f walked %.0f rows listed %.1f" % (
                        lo, hi, int(sel.sum()), g[sel, 5].mean(), g[sel, 0].mean(), g[sel, 1].mean(), g[sel, 2].mean(), g[sel, 3].mean(), g[sel, 4].mean()))
        lf = life[ok]
        print("   wave life (us): mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f; sum over waves / (CUs x 1 us): %.1f" % (
            lf.mean(), np.percentile(lf, 50), np.percentile(lf, 90), np.percentile(lf, 99), lf.max(), lf.sum() / 256.0))
        # which waves end last
        order = np.argsort(-en)[:12]
        print("   last to end:   " + " ".join("w%d[%.0f-%.0f]" % (w, st[w], en[w]) for w in order))
    tot = wv.sum(1)
    print("   wave total: mean %.1f  p99 %.1f  max %.1f us; waves over 4x the mean: %d" % (tot.mean(), np.percentile(tot, 99), tot.max(), int((tot > 4 * tot.mean()).sum())))
    print("mode %d: fwd %.1f us" % (mode, a.elapsed_time(b) / 20 * 1e3))
    ph = ph[ph[:, 0] > 0]
    t0 = ph[:, 0].min()
    print("   kernel end (last stamped wg) +%.1f us" % ((int(ph[:, 9].max()) - int(t0)) / 100.0))
    for w in range(len(ph)):
        d = (ph[w, 1:10].astype(np.int64) - ph[w, 0:9].astype(np.int64)) / 100.0
        print("   wg %4d start +%.2f us: " % (w * 512, (int(ph[w, 0]) - int(t0)) / 100.0) + " ".join("p%d %.2f" % (k, v) for k, v in enumerate(d)) +
              "  total %.2f" % ((int(ph[w, 9]) - int(ph[w, 0])) / 100.0))
