#!/bin/bash
# tools/libpp_hip_<tag>.so: the shipped library with chamfer_grid.hip recompiled with extra flags (A/B of code paths
# on the GPU box: PP_LIB=tools/libpp_hip_<tag>.so python tools/tile_modes.py ...).  usage: build_variant_lib.sh <tag> <flags...>
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fPIC -std=c++17 -Wall -Wno-unused-function \
  "$@" -Iinclude -Ipytorch_points_amd/csrc -c pytorch_points_amd/csrc/chamfer_grid.hip -o /tmp/chamfer_grid_$tag.o 2>/dev/null
objs=$(ls pytorch_points_amd/build/*.o | grep -v chamfer_grid.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/chamfer_grid_$tag.o -o tools/libpp_hip_$tag.so
