#!/bin/bash
python tools/build_phases.py sphere sphere > gpurun_out/build_phases3.log 2>&1
cat gpurun_out/build_phases3.log
