#!/bin/bash
python tools/build_phases.py sphere two_scales blobs8 gaussian > gpurun_out/build_phases.log 2>&1
cat gpurun_out/build_phases.log
