#!/bin/bash
# round 5: the backward (both batches loaded before the barrier, own loads first, 16-byte write-out)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chamfer.py tests/test_gpu_deterministic.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.txt
timeout 120 ./tools/bwd_probe 2>&1 | grep -v amdgpu.ids > $O/bwd_probe.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
cat $O/tests.txt $O/bwd_probe.txt
python3 -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['fwd_ms'], d['launch_modes_ms_per_step']); print(d['roofline']['build_kernel_ms'], d['roofline']['stage_a_kernel_ms'], d['roofline']['rest_kernel_ms'])"
