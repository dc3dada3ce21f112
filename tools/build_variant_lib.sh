#!/bin/bash
# tools/libpp_hip_<tag>.so: the shipped library with one source (SRC=<name>, default chamfer_grid) recompiled with extra flags (A/B of code paths
# on the GPU box: PP_LIB=tools/libpp_hip_<tag>.so python tools/tile_modes.py ...).  usage: build_variant_lib.sh <tag> <flags...>
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
src=${SRC:-chamfer_grid}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fPIC -std=c++17 -Wall -Wno-unused-function \
  "$@" -Iinclude -Ipytorch_points_amd/csrc -c pytorch_points_amd/csrc/$src.hip -o /tmp/${src}_$tag.o 2>/dev/null
objs=$(ls pytorch_points_amd/build/*.o | grep -v /$src.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/${src}_$tag.o -o tools/libpp_hip_$tag.so
