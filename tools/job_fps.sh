#!/bin/bash
# FPS cluster kernel: phase probe per poll variant, parity tests, config-3 bench line
for v in 0 1 2 3; do timeout 120 ./tools/fps_probe $v; done > gpurun_out/fps_variants.log 2>&1
timeout 900 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_golden.py -m gpu -x -q -k "fps or furthest or FPS" > gpurun_out/pytest_fps.log 2>&1
timeout 300 python bench.py --workload fps --steps 5 --warmup 2 > gpurun_out/bench_fps.json 2> gpurun_out/bench_fps.err
tail -3 gpurun_out/pytest_fps.log; cat gpurun_out/fps_variants.log; cat gpurun_out/bench_fps.json
