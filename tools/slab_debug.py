"""where the fused kernel's results differ from the every-pair kernel's (a debugging aid)"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import losses
dev = torch.device("cuda:0")
B, N = 32, 16384
L = _lib.lib()
sett = L.pp_debug_set_nmdistance_tile; sett.argtypes = [ctypes.c_int]; sett.restype = None
sets = L.pp_debug_set_nmdistance_search; sets.argtypes = [ctypes.c_int]; sets.restype = None
kind = sys.argv[1] if len(sys.argv) > 1 else "same"
x1 = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev)
x2 = torch.from_numpy(S.unit_sphere(0 if kind == "same" else 1, B, N)).to(dev)
def outs():
    return (torch.empty(B, N, device=dev), torch.empty(B, N, device=dev),
            torch.empty(B, N, dtype=torch.int32, device=dev), torch.empty(B, N, dtype=torch.int32, device=dev))
sets(1); ref = outs(); losses.nmdistance_forward(x1, x2, *ref); sets(0)
sett(-2)
for rep in range(4):
    o = outs()
    for t in o: t.fill_(-7)
    losses.nmdistance_forward(x1, x2, *o)
    wd = (ctypes.c_uint * (8 * B))()
    wsb = _lib.cached_workspaces("nmdistance")[0]
    fn = L.pp_debug_nmdistance_slab_state; fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]; fn.restype = ctypes.c_int
    fn(wsb.data_ptr(), B, N, N, wd)
    st = np.array(list(wd)).reshape(B, 8)
    bad1 = (o[0] != ref[0]) | (o[2] != ref[2]); bad2 = (o[1] != ref[1]) | (o[3] != ref[3])
    print("rep", rep, "declined elements", np.nonzero((st != 0).any(1))[0].tolist(), "mismatches", int(bad1.sum()), int(bad2.sum()))
    for b in range(B):
        n1, n2 = int(bad1[b].sum()), int(bad2[b].sum())
        if n1 + n2:
            z = x1[b, :, 2].cpu().numpy(); zmin = min(x1[b, :, 2].min().item(), x2[b, :, 2].min().item()); ext = 2.0
            ii = np.nonzero(bad1[b].cpu().numpy())[0]
            layers = np.floor((z[ii] - zmin) / (ext / 32 * 1.0001)).astype(int)
            print("  b", b, "state", st[b].tolist(), "bad", n1, n2, "layers of bad dir-0 queries", np.bincount(np.clip(layers, 0, 31), minlength=32).tolist())
            k = ii[:3]
            print("    sample", [(int(i), float(o[0][b, i]), float(ref[0][b, i]), int(o[2][b, i]), int(ref[2][b, i])) for i in k])
sett(0)
