#!/bin/bash
python tools/fps_variants.py > gpurun_out/fps_variants2.log 2>&1
timeout 900 python -m pytest tests/test_gpu_chamfer.py -m gpu -x -q -k "double or rejects or python_classes or autograd" > gpurun_out/pytest15.log 2>&1
cat gpurun_out/fps_variants2.log; tail -15 gpurun_out/pytest15.log
