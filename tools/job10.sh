#!/bin/bash
python tools/tile_modes.py sphere cube gaussian plane 2>&1 | cut -c1-200 | tee gpurun_out/tile_modes10.log
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/pytest_all10.log
python bench.py --steps 300 > gpurun_out/bench10.json 2> gpurun_out/bench10.err
