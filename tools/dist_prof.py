import sys, numpy as np, torch
sys.path.insert(0, ".")
from pytorch_points_amd._ext import losses
dev = torch.device("cuda:0")
B, N = 32, 16384
rng = np.random.default_rng(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "cube"
def mk():
    if kind == "cube": return rng.random((B, N, 3), dtype=np.float32)
    if kind == "gaussian": return rng.standard_normal((B, N, 3)).astype(np.float32)
    if kind == "blobs8":
        c = rng.random((B, 8, 3), dtype=np.float32) * 2
        return (c[:, rng.integers(0, 8, N)] + rng.standard_normal((B, N, 3)).astype(np.float32) * 0.02).astype(np.float32)
    if kind == "shapenet_like":
        x = rng.random((B, N, 3), dtype=np.float32) - 0.5
        q = N // 4
        x[:, :q, 2] = -0.5; x[:, q:2 * q, 0] = 0.2
        th = rng.random((B, q)) * 6.283; x[:, 2 * q:3 * q, 0] = 0.3 * np.cos(th); x[:, 2 * q:3 * q, 1] = 0.3 * np.sin(th)
        return x.astype(np.float32)
    x = np.zeros((B, N, 3), np.float32); x[..., 0] = rng.random((B, N), dtype=np.float32); return x
x1 = torch.from_numpy(mk()).to(dev); x2 = torch.from_numpy(mk()).to(dev)
d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
for _ in range(6): losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
torch.cuda.synchronize()
ws = list(losses._nmd_workspace.values())[0]
S2 = 2 * B
cnt = ws[64 * S2: 64 * S2 + 4 * 2 * S2].view(torch.int32).cpu().numpy()
sets = ws[:64 * S2].view(torch.float32).cpu().numpy().reshape(S2, 16)
seti = ws[:64 * S2].view(torch.int32).cpu().numpy().reshape(S2, 16)
print(kind, "unresolved after stage A per set (mean): %.0f  -> brute-force list per set (mean): %.0f  of %d" % (cnt[S2:].mean(), cnt[:S2].mean(), N))
print("   set 0: h=%.4f grid %dx%dx%d useless=%d crowd=%s" % (sets[0, 3], seti[0, 5], seti[0, 6], seti[0, 7], seti[0, 8], seti[0, 12:16]))
