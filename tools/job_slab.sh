#!/bin/bash
# chamfer_slab.hip: the phases of the fused kernel (variants that leave after phase n) and the whole, sphere clouds of config 2
for n in 9 1 2 3 4; do
  [ -f tools/libpp_hip_slab$n.so ] && { echo "stop after phase $n (results unwritten: MISMATCH expected)"; PP_LIB=tools/libpp_hip_slab$n.so PP_TILE_MODES=-2 timeout 120 python tools/tile_modes.py sphere 2>&1 | tail -1; }
done
PP_TILE_MODES=512,-2 timeout 250 python tools/tile_modes.py ${KINDS:-sphere same} 2>&1 | tail -12
