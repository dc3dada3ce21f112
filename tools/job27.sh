#!/bin/bash
python tools/tile_modes.py two_scales sphere gaussian blobs8 cube 2>&1 | cut -c1-215 > gpurun_out/tile_modes27.log
timeout 2400 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_chamfer.py tests/test_gpu_nonfinite.py tests/test_gpu_fuzz.py tests/test_gpu_sampling.py tests/test_gpu_knn.py -m gpu -x -q > gpurun_out/pytest27.log 2>&1
cat gpurun_out/tile_modes27.log; tail -4 gpurun_out/pytest27.log
