#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for k in shapenet_like gaussian two_scales; do echo "== $k"; PP_PROBE_KIND=$k timeout 300 python3 tools/query_probe.py 512 2>&1 | grep -v amdgpu.ids | grep "stage A inside\|per wave\|mode"; done
