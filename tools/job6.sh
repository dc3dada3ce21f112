#!/bin/bash
PP_PROBE_LIB=libpp_hip_probe_a.so python tools/query_probe.py 512 513 > gpurun_out/qprobe6.log 2>&1
