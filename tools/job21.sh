#!/bin/bash
timeout 600 python tools/time_misc_ops.py > gpurun_out/misc21.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_golden.py -m gpu -x -q -k "three_nn or interpolate" > gpurun_out/pytest21.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_chamfer_grid.py -m gpu -x -q -k "far_clouds or clustered or other_distributions" > gpurun_out/pytest21b.log 2>&1
head -12 gpurun_out/misc21.log; tail -3 gpurun_out/pytest21.log; tail -3 gpurun_out/pytest21b.log
