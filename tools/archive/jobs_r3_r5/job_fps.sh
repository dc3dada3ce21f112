#!/bin/bash
# FPS: parity tests of the kernels, the round probe, timings over shapes / clouds, config-3 bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_golden.py tests/test_gpu_nonfinite.py tests/test_gpu_fuzz.py -m gpu -x -q -k "fps or furthest or FPS" > gpurun_out/pytest_fps.log 2>&1
tail -15 gpurun_out/pytest_fps.log
(timeout 120 ./tools/fps_bucket_probe 16 65536 4096; PP_PROBE_CHAIN=1 timeout 120 ./tools/fps_bucket_probe 16 65536 4096) > gpurun_out/fps_bucket_probe.txt 2>&1
cat gpurun_out/fps_bucket_probe.txt
timeout 600 python tools/fps_time.py > gpurun_out/fps_time.txt 2>&1
cat gpurun_out/fps_time.txt
timeout 300 python bench.py --workload fps --steps 5 --warmup 2 > gpurun_out/bench_fps.json 2> gpurun_out/bench_fps.err
cat gpurun_out/bench_fps.json
