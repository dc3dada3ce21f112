#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out/fulltest
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/fulltest/tests.txt
