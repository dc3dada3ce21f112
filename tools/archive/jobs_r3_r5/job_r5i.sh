#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
timeout 300 python tools/knn_degenerate_time.py 2>&1 | grep -v amdgpu.ids | tee $O/knn_degenerate.txt
