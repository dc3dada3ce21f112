#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_fuzz.py tests/test_gpu_chamfer.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -3
PP_FUZZ_SEEDS=400 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "chamfer or config2 or nmdist" 2>&1 | tail -3
