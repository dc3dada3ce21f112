#!/bin/bash
python tools/tile_modes.py gaussian blobs8 disjoint two_scales shapenet_like cube sphere 2>&1 | cut -c1-215 > gpurun_out/tile_modes24.log
timeout 600 python tools/time_misc_ops.py 2>&1 | grep -i "labeled\|knn\|three_nn" > gpurun_out/misc24.log
timeout 2400 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_chamfer.py tests/test_gpu_nonfinite.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/pytest24.log 2>&1
cat gpurun_out/tile_modes24.log gpurun_out/misc24.log; tail -4 gpurun_out/pytest24.log
