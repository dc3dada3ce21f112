#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
PP_TILE_MODES=512 timeout 600 python3 tools/tile_modes.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
