#!/bin/bash
python tools/tile_modes.py 2>&1 | tee gpurun_out/tile_modes4.log
PROF_LINES=8 bash tools/prof.sh t512 tools/fwd_loop.py 512 sphere 100 > /dev/null
python3 - <<'PY' | tee gpurun_out/prof_t512.txt
import csv,glob
f=glob.glob('gpurun_out/prof_t512/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-60s calls %5s avg %8.1f us  min %8.1f max %8.1f" % (r["Name"].replace("void (anonymous namespace)::","")[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
timeout 900 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_chamfer.py tests/test_gpu_golden.py tests/test_gpu_nonfinite.py -x -q 2>&1 | tail -5 | tee gpurun_out/pytest_chamfer4.log
