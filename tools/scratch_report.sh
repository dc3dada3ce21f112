#!/bin/bash
# usage: tools/scratch_report.sh <file.hip> : scratch instructions per function in the gfx950 ISA of one source
# (spills on a main path show up here; -Rpass-analysis only gives the per-kernel total)
set -e
src=$1
out=/tmp/$(basename "$src").s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -std=c++17 -I/root/repo/include \
  -I/root/repo/pytorch_points_amd/csrc -S --cuda-device-only "$src" -o "$out" 2>/dev/null
python3 - "$out" <<'PY'
import re, sys
cur = None
cnt = {}
regs = {}
for ln in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', ln)
    if m:
        cur = m.group(1)
    if cur and ('scratch_store' in ln or 'scratch_load' in ln):
        cnt[cur] = cnt.get(cur, 0) + 1
    m = re.match(r'\s*\.vgpr_count:\s*(\d+)', ln)
    if m:
        regs['last'] = m.group(1)
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(v, k[:100])
if not cnt:
    print("no scratch instructions")
PY
