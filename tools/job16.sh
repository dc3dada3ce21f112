#!/bin/bash
for k in disjoint blobs8 gaussian two_scales; do echo "== $k"; PP_PROBE_KIND=$k timeout 300 python tools/query_probe.py -1 0 2>&1 | grep -v "^   wg " ; done > gpurun_out/qprobe_far.log 2>&1
cat gpurun_out/qprobe_far.log
