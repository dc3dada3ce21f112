#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_fuzz.py tests/test_gpu_chamfer.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for v in "" base; do echo "== variant '$v'"
  if [ -n "$v" ]; then export PP_LIB=tools/libpp_hip_$v.so; else unset PP_LIB; fi
  PP_TILE_MODES=512 timeout 600 python3 tools/tile_modes.py sphere cube gaussian shapenet_like two_scales blobs8 disjoint 2>&1 | grep -v amdgpu.ids
done; done
unset PP_LIB; timeout 300 python3 tools/labeled_time.py 2>&1 | grep -v amdgpu.ids | head -4
