#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 2500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for v in "" noregs; do echo "== variant '$v'"
  if [ -n "$v" ]; then export PP_LIB=tools/libpp_hip_$v.so; else unset PP_LIB; fi
  PP_TILE_MODES=512 timeout 600 python3 tools/tile_modes.py sphere gaussian two_scales blobs8 line 2>&1 | grep -v amdgpu.ids
done
unset PP_LIB; timeout 300 python3 tools/dist_probe.py 2>&1 | grep -v amdgpu.ids | grep "blobs8\|two_scales"
