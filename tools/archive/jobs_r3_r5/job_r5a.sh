#!/bin/bash
# round 5, first GPU job: the Chamfer / shard tests after the clean-up, the build's phase clocks, a bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r5a
timeout 1500 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_shard.py tests/test_gpu_chamfer.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r5a/tests.txt
PP_PROBE_LIB=libpp_hip_bprobe.so timeout 300 python tools/build_phases.py sphere gaussian 2>&1 | grep -v amdgpu.ids > gpurun_out/r5a/build_phases.txt
timeout 600 python bench.py > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err
tail -c 600 gpurun_out/r5a/bench.err
cat gpurun_out/r5a/tests.txt gpurun_out/r5a/build_phases.txt
