#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
export PP_TILE_MODES=512
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cube -- python3 tools/tile_modes.py cube > /tmp/cube.log 2>&1
grep -v amdgpu.ids /tmp/cube.log | grep "cube"
f=$(find /tmp/prof_cube -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-220
