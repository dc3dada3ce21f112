#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sampling.py -x -q -m gpu -k "accumulating_and_overwriting" 2>&1 | tail -5
