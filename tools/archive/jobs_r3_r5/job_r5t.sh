#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for i in 1 2 3; do
echo "-- LDS-sorted build"; timeout 100 python tools/bq_time.py 2>&1 | grep -v amdgpu.ids | tail -1
echo "-- general build"; PP_LIB=tools/libpp_hip_bqgen.so timeout 100 python tools/bq_time.py 2>&1 | grep -v amdgpu.ids | tail -1
done
