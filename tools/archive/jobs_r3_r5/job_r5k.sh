#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r5k; mkdir -p $O
echo "== LDS-sorted build"; bash tools/pmc_multi.sh bf "WRITE_SIZE" "FETCH_SIZE" -- tools/fwd_loop.py 0 sphere 20 2>&1 | grep -v amdgpu.ids | grep -A3 "grid_build_kernel"
PP_TILE_MODES=512 timeout 300 python3 tools/tile_modes.py sphere gaussian 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_sampling.py tests/test_gpu_knn.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --workload ball_group --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print({k:d[k] for k in d if k.endswith('_ms')})"
