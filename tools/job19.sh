#!/bin/bash
for v in "" w5 w4; do
  echo "== variant '$v'"
  if [ -z "$v" ]; then python tools/tile_modes.py gaussian blobs8 disjoint two_scales shapenet_like cube 2>&1 | cut -c1-215
  else PP_LIB=tools/libpp_hip_$v.so python tools/tile_modes.py gaussian blobs8 disjoint two_scales shapenet_like cube 2>&1 | cut -c1-215; fi
done > gpurun_out/variants19.log 2>&1
cat gpurun_out/variants19.log
