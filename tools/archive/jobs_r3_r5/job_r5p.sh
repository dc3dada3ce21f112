#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
PP_PROBE_LIB=libpp_hip_aprobe.so PP_PROBE_KIND=sphere timeout 300 python tools/query_probe.py 512 2>&1 | grep -v amdgpu.ids | cut -c1-600 | head -14
