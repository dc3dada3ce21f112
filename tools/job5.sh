#!/bin/bash
python tools/tile_modes.py sphere same cube 2>&1 | tee gpurun_out/tile_modes5.log
PROF_LINES=8 bash tools/prof.sh t512 tools/fwd_loop.py 512 sphere 100 > /dev/null
python3 - <<'PY' | tee gpurun_out/prof_t512.txt
import csv,glob
f=glob.glob('gpurun_out/prof_t512/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-60s calls %5s avg %8.1f us  min %8.1f max %8.1f" % (r["Name"].replace("void (anonymous namespace)::","")[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
python tools/query_probe.py 512 > gpurun_out/qprobe5.log 2>&1
timeout 600 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_chamfer.py -x -q 2>&1 | tail -3 | tee gpurun_out/pytest5.log
PP_PROBE_LIB=libpp_hip_probe_a.so python tools/query_probe.py 512 > gpurun_out/qprobe6.log 2>&1
