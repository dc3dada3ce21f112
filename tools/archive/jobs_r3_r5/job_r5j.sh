#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r5j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_knn.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.txt
timeout 300 python bench.py --workload ball_group --steps 20 --warmup 5 > $O/bench_bg.json 2> $O/bench_bg.err
python3 -c "
import json;d=json.loads(open('$O/bench_bg.json').read().strip().splitlines()[-1])
print({k:d[k] for k in d if k.endswith('_ms')})"
timeout 600 python tools/time_misc_ops.py 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/misc.txt
