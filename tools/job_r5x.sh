#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for rep in 1 2; do
for v in "" noball; do echo "== variant '$v'"
  if [ -n "$v" ]; then export PP_LIB=tools/libpp_hip_$v.so; else unset PP_LIB; fi
  PP_TILE_MODES=512 timeout 600 python3 tools/tile_modes.py blobs8 gaussian two_scales 2>&1 | grep -v amdgpu.ids
done; done
