#!/bin/bash
python tools/odd_size_timing.py > gpurun_out/odd26.log 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest26.log 2>&1
cat gpurun_out/odd26.log; tail -4 gpurun_out/pytest26.log
