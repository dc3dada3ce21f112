#!/bin/bash
python tools/tile_modes.py blobs8 disjoint gaussian sphere 2>&1 | cut -c1-150 | tee gpurun_out/tile_modes11.log
timeout 1200 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_chamfer.py tests/test_gpu_fuzz.py tests/test_gpu_nonfinite.py tests/test_gpu_shard.py -x -q 2>&1 | tail -3 | tee gpurun_out/pytest11.log
bash tools/job13.sh
