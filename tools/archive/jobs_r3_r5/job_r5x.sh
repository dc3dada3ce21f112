#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 300 python3 tools/host_step_probe.py 2>&1 | grep -v amdgpu.ids | tail -8
