#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_chamfer_grid.py -x -q -m gpu -k "ball_stage" 2>&1 | tail -4
