#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 200 python tools/fps_skew_time.py 2>&1 | grep -v amdgpu.ids | tail -1
for s in 1 3 5 7; do PP_LIB=tools/libpp_hip_fskew$s.so timeout 200 python tools/fps_skew_time.py 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout 200 python tools/fps_skew_time.py 2>&1 | grep -v amdgpu.ids | tail -1
