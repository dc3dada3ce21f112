#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 2500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
