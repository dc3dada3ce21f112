#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_chamfer_grid.py -x -q -m gpu 2>&1 | grep -v "^  File" | tail -8
echo "== LDS-sorted build"; bash tools/pmc_multi.sh bf "WRITE_SIZE" -- tools/fwd_loop.py 0 sphere 20 2>&1 | grep -v amdgpu.ids | grep -A2 "grid_build_kernel"
PP_TILE_MODES=512 timeout 300 python3 tools/tile_modes.py sphere gaussian 2>&1 | grep -v amdgpu.ids
