#!/bin/bash
PP_PROBE_KIND=disjoint python tools/query_probe.py 512 > gpurun_out/qprobe_disjoint.log 2>&1
PP_PROBE_KIND=blobs8 python tools/query_probe.py 512 > gpurun_out/qprobe_blobs8.log 2>&1
