#!/bin/bash
python tools/tile_modes.py same sphere 2>&1 | tee gpurun_out/tile_modes7.log
python3 - <<'PY' > gpurun_out/fwd_same.py
PY
cat > /tmp/fwd_same.py <<'PY'
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
from pytorch_points_amd import _lib, synthetic as S
from pytorch_points_amd._ext import losses
dev = torch.device("cuda:0")
B, N = 32, 16384
x1 = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); x2 = x1.clone()
d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, N, device=dev)
i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, N, dtype=torch.int32, device=dev)
for _ in range(100):
    losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
torch.cuda.synchronize()
PY
cp /tmp/fwd_same.py gpurun_out/fwd_same.py
PROF_LINES=8 bash tools/prof.sh same gpurun_out/fwd_same.py > /dev/null
python3 - <<'PY' | tee gpurun_out/prof_same.txt
import csv,glob
f=glob.glob('gpurun_out/prof_same/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-60s calls %5s avg %8.1f us  min %8.1f max %8.1f" % (r["Name"].replace("void (anonymous namespace)::","")[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
