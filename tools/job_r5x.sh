#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for v in "" r30 r40 r60; do echo "== variant '$v'"
  if [ -n "$v" ]; then export PP_LIB=tools/libpp_hip_$v.so; else unset PP_LIB; fi
  PP_TILE_MODES=512 timeout 600 python3 tools/tile_modes.py cube gaussian shapenet_like two_scales blobs8 disjoint 2>&1 | grep -v amdgpu.ids
done
