#!/bin/bash
# round 5: the LDS-sorted build: correctness (Chamfer suites), phase clocks, forward per distribution, a bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_chamfer.py tests/test_gpu_golden.py tests/test_gpu_nonfinite.py -x -q -m gpu 2>&1 | tail -15 > $O/tests.txt
PP_PROBE_LIB=libpp_hip_bprobe.so timeout 300 python tools/build_phases.py sphere gaussian blobs8 2>&1 | grep -v amdgpu.ids > $O/build_phases.txt
PP_TILE_MODES=512 timeout 600 python3 tools/tile_modes.py 2>&1 | grep -v amdgpu.ids > $O/dists.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.err
cat $O/tests.txt $O/build_phases.txt $O/dists.txt
python3 -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['fwd_ms'], d['launch_modes_ms_per_step']); print(d['roofline']['build_kernel_ms'], d['roofline']['stage_a_kernel_ms'], d['roofline']['rest_kernel_ms'])
print(d.get('other_distributions_fwd_ms'))"
