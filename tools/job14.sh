#!/bin/bash
python tools/tile_modes.py gaussian blobs8 disjoint two_scales shapenet_like sphere 2>&1 | cut -c1-150 | tee gpurun_out/tile_modes14.log
python bench.py --steps 300 --no-extras --no-cpu-baseline > gpurun_out/bench14.json 2> gpurun_out/bench14.err
python3 -c "
import json
d=json.loads(open('gpurun_out/bench14.json').read().strip().split('\n')[-1])
print('ms_per_step', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_events_median'],4), 'fwd', round(d['fwd_ms'],4), d['launch_modes_ms_per_step'])
" | tee gpurun_out/bench14.txt
timeout 1200 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_chamfer.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3 | tee gpurun_out/pytest14.log
