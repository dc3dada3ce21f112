#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 300 python3 tools/build_phases.py sphere two_scales blobs8 gaussian shapenet_like 2>&1 | grep -v amdgpu.ids
