#!/bin/bash
for k in disjoint; do echo "== $k"; PP_PROBE_KIND=$k timeout 300 python tools/query_probe.py 0 2>&1; done > gpurun_out/qprobe_far3.log 2>&1
cat gpurun_out/qprobe_far3.log
