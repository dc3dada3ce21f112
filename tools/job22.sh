#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_deterministic.py -m gpu -x -q > gpurun_out/pytest22.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_chamfer_grid.py -m gpu -x -q -k "far_clouds or clustered or other_distributions" > gpurun_out/pytest22b.log 2>&1
tail -12 gpurun_out/pytest22.log; tail -12 gpurun_out/pytest22b.log
