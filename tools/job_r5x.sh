#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 2500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 300 python3 tools/dist_probe.py 2>&1 | grep -v amdgpu.ids
