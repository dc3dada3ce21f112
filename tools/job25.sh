#!/bin/bash
python tools/tile_modes.py gaussian blobs8 disjoint two_scales 2>&1 | cut -c1-215 > gpurun_out/tile_modes25.log
timeout 2400 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_nonfinite.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/pytest25.log 2>&1
cat gpurun_out/tile_modes25.log; tail -4 gpurun_out/pytest25.log
