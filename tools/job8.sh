#!/bin/bash
python tools/tile_modes.py sphere 2>&1 | tee gpurun_out/tile_modes8.log
python bench.py --steps 300 --no-extras --no-cpu-baseline > gpurun_out/bench8_512.json 2> gpurun_out/bench8_512.err
PP_NMDISTANCE_TILE=1024 python bench.py --steps 300 --no-extras --no-cpu-baseline > gpurun_out/bench8_1024.json 2> gpurun_out/bench8_1024.err
PP_NMDISTANCE_TILE=-1 python bench.py --steps 300 --no-extras --no-cpu-baseline > gpurun_out/bench8_old.json 2> gpurun_out/bench8_old.err
timeout 900 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_shard.py -x -q 2>&1 | tail -4 | tee gpurun_out/pytest8.log
