#!/bin/bash
# round 5: timeline of the list kernel's waves on the far-field clouds
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r5d; mkdir -p $O
for k in gaussian blobs8 shapenet_like two_scales; do
  echo "== $k"; PP_PROBE_LIB=libpp_hip_qprobe.so PP_PROBE_KIND=$k timeout 300 python tools/query_probe.py 512 2>&1 | grep -v amdgpu.ids | grep -v "^   wg "
done > $O/timeline.txt
cat $O/timeline.txt
