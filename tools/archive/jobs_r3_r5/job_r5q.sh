#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for i in 1 2 3; do
  echo "-- with prefetch"; PP_TILE_MODES=512 timeout 200 python3 tools/tile_modes.py sphere 2>&1 | grep -v amdgpu.ids
  echo "-- without";       PP_LIB=tools/libpp_hip_noqb.so PP_TILE_MODES=512 timeout 200 python3 tools/tile_modes.py sphere 2>&1 | grep -v amdgpu.ids
done
