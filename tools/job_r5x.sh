#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
PP_NMDISTANCE_TILE=256 timeout 1500 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_fuzz.py tests/test_gpu_chamfer.py tests/test_gpu_golden.py -q -m gpu --deselect tests/test_gpu_chamfer_grid.py::test_stage_a_serves_an_evenly_sampled_surface 2>&1 | tail -3
