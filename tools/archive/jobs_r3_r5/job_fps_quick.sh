#!/bin/bash
# FPS, quick loop: the bucket kernel's parity tests, the round probe, config-3 timings
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -m gpu -x -q -k "fps or furthest or FPS" > gpurun_out/pytest_fps.log 2>&1
tail -5 gpurun_out/pytest_fps.log
(timeout 120 ./tools/fps_bucket_probe 16 65536 4096) > gpurun_out/fps_bucket_probe.txt 2>&1
head -8 gpurun_out/fps_bucket_probe.txt; tail -3 gpurun_out/fps_bucket_probe.txt
timeout 600 python tools/fps_time.py quick > gpurun_out/fps_time.txt 2>&1
cat gpurun_out/fps_time.txt
