#!/bin/bash
# the batch-sharded exchange: one-rank RCCL paths, the distributed bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_shard.py tests/test_gpu_bench_contract.py -m gpu -x -q > gpurun_out/pytest_shard.log 2>&1
tail -25 gpurun_out/pytest_shard.log
