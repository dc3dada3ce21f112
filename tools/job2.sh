#!/bin/bash
python tools/tile_modes.py 2>&1 | tee gpurun_out/tile_modes2.log
python tools/query_probe.py 512 1024 > gpurun_out/qprobe2.log 2>&1
timeout 900 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_chamfer.py -x -q 2>&1 | tail -5 | tee gpurun_out/pytest_chamfer2.log
