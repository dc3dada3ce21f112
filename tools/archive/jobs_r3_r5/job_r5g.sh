#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -30
import numpy as np, torch, sys
sys.path.insert(0, ".")
from pytorch_points_amd import synthetic as S
from pytorch_points_amd._ext import sampling
cuda = torch.device("cuda:0")
b, c, n, npoint, ns, r = 2, 6, 4096, 512, 64, 0.08
x = S.unit_sphere(45, b, n)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
idx = sampling.ball_query(t(x[:, :npoint]), t(x), r, ns)
idx[:, 0, :] = n - 1
idx[:, -1, 1::2] = idx[:, -1, 0:1]
go = t(S.normal(46, (b, c, npoint, ns)))
got = sampling.group_points_grad(go, idx, n)
ref = torch.zeros(b, c, n, device=cuda, dtype=torch.float64)
ref.scatter_add_(2, idx.long().reshape(b, 1, -1).expand(-1, c, -1), go.double().reshape(b, c, -1))
d = (got.double() - ref).abs()
print("max err", float(d.max()), "bad elements", int((d > 1e-4).sum()))
bad = torch.nonzero(d > 1e-4)[:10]
print(bad.tolist())
for bb, cc, nn in bad[:5].tolist():
    pos = torch.nonzero(idx[bb].reshape(-1) == nn).reshape(-1)
    print("dest", nn, "got", float(got[bb, cc, nn]), "ref", float(ref[bb, cc, nn]), "positions", pos[:40].tolist(), "count", len(pos))
PY
bash tools/job_r5f.sh
