#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_deterministic.py -m gpu -x -q -k "fps or ordered or deterministic" > gpurun_out/pytest23.log 2>&1
tail -12 gpurun_out/pytest23.log
