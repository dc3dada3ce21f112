#!/bin/bash
python tools/tile_modes.py disjoint blobs8 gaussian two_scales shapenet_like sphere cube > gpurun_out/tile_modes17.log 2>&1
for k in disjoint blobs8 gaussian; do echo "== $k"; PP_PROBE_KIND=$k timeout 300 python tools/query_probe.py 0 2>&1 | grep -v "^   wg " ; done > gpurun_out/qprobe_far2.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_fuzz.py tests/test_gpu_nonfinite.py -m gpu -x -q > gpurun_out/pytest17.log 2>&1
cat gpurun_out/tile_modes17.log gpurun_out/qprobe_far2.log; tail -4 gpurun_out/pytest17.log
