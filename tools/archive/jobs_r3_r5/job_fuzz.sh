#!/bin/bash
PP_FUZZ_SEEDS=1500 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/pytest_fuzz.log 2>&1
tail -5 gpurun_out/pytest_fuzz.log
