#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for v in "" nolb lb15 lb05; do echo "== variant '$v'"
  if [ -n "$v" ]; then export PP_LIB=tools/libpp_hip_$v.so; else unset PP_LIB; fi
  PP_TILE_MODES=512 timeout 600 python3 tools/tile_modes.py two_scales blobs8 line 2>&1 | grep -v amdgpu.ids
done
