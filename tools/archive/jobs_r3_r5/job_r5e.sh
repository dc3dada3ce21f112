#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
PP_TILE_MODES=512 timeout 600 python3 tools/tile_modes.py 2>&1 | grep -v amdgpu.ids > $O/dists.txt
cat $O/dists.txt
for k in gaussian shapenet_like; do
  echo "== $k"; PP_PROBE_LIB=libpp_hip_qprobe.so PP_PROBE_KIND=$k timeout 300 python tools/query_probe.py 512 2>&1 | grep -v amdgpu.ids | grep -v "^   wg " | cut -c1-700
done > $O/timeline.txt
cat $O/timeline.txt
timeout 900 python -m pytest tests/test_gpu_chamfer_grid.py -x -q -m gpu 2>&1 | tail -3
