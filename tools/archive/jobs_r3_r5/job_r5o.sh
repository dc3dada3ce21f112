#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_chamfer.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['fwd_ms'], d['launch_modes_ms_per_step']['ext']); r=d['roofline']; print(r['build_kernel_ms'], r['stage_a_kernel_ms'], r['rest_kernel_ms'], r['backward_kernel_ms'])"; done
