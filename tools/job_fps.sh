#!/bin/bash
# FPS cluster kernel: phase probe, parity tests, config-3 bench line
timeout 120 ./tools/fps_probe > gpurun_out/fps_probe.log 2>&1
timeout 900 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_golden.py -m gpu -x -q -k "fps or furthest or FPS" > gpurun_out/pytest_fps.log 2>&1
timeout 300 python bench.py --workload fps --steps 5 --warmup 2 > gpurun_out/bench_fps.json 2> gpurun_out/bench_fps.err
tail -3 gpurun_out/pytest_fps.log; cat gpurun_out/fps_probe.log; cat gpurun_out/bench_fps.json
