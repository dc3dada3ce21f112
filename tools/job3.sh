#!/bin/bash
python tools/tile_modes.py sphere cube gaussian plane shapenet_like 2>&1 | tee gpurun_out/tile_modes3.log
python tools/query_probe.py 512 > gpurun_out/qprobe3.log 2>&1
