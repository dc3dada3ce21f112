#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for l in libpp_hip_bprobe.so libpp_hip_bprobe_masked.so libpp_hip_bprobe.so libpp_hip_bprobe_masked.so; do echo "== $l"; PP_PROBE_LIB=$l timeout 200 python tools/build_phases.py sphere 2>&1 | grep -v amdgpu.ids | head -2; done
