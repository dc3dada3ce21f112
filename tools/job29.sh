#!/bin/bash
python tools/build_phases.py blobs8 two_scales > gpurun_out/build_phases2.log 2>&1
python tools/tile_modes.py two_scales blobs8 disjoint gaussian 2>&1 | cut -c1-215 > gpurun_out/tile_modes29.log
timeout 2400 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_chamfer.py tests/test_gpu_nonfinite.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/pytest29.log 2>&1
cat gpurun_out/build_phases2.log gpurun_out/tile_modes29.log; tail -4 gpurun_out/pytest29.log
