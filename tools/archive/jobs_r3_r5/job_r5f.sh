#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r5f; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_deterministic.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.txt
timeout 300 python bench.py --workload ball_group --with-backward --steps 20 --warmup 5 > $O/bench_bg.json 2> $O/bench_bg.err
python3 -c "
import json;d=json.loads(open('$O/bench_bg.json').read().strip().splitlines()[-1])
print({k:d[k] for k in d if k.endswith('_ms')})"
timeout 300 python tools/gpg_time.py 2>&1 | grep -v amdgpu.ids | tail -8
