#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -x -q -m gpu -k "ball or query or fuzz or golden" 2>&1 | tail -3
PP_LIB=tools/libpp_hip_bqprobe.so timeout 300 python tools/bq_phases.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do timeout 300 python bench.py --workload ball_group --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print({k:d[k] for k in d if k.endswith('_ms')})"; done
