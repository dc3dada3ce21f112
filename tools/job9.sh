#!/bin/bash
python tools/tile_modes.py sphere cube gaussian blobs8 two_scales plane 2>&1 | cut -c1-300 | tee gpurun_out/tile_modes9.log
timeout 900 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_chamfer.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3 | tee gpurun_out/pytest9.log
