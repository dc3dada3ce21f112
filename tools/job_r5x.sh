#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 2000 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_knn.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -2
echo "== as built"; timeout 300 python3 tools/knn_time.py 2>&1 | grep -v amdgpu.ids
echo "== knn box"; PP_LIB=tools/libpp_hip_knn_box.so timeout 300 python3 tools/knn_time.py 2>&1 | grep -v amdgpu.ids
echo "== as built"; timeout 300 python3 tools/knn_time.py 2>&1 | grep -v amdgpu.ids
