#!/bin/bash
for k in blobs8 gaussian; do echo "== $k"; PP_PROBE_KIND=$k timeout 300 python tools/query_probe.py 0 2>&1 | grep -v "^   wg"; done > gpurun_out/qprobe_far4.log 2>&1
cat gpurun_out/qprobe_far4.log
