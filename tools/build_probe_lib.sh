#!/bin/bash
# tools/libpp_hip_probe.so (or $PP_PROBE_OUT): the library with chamfer_grid.hip compiled -DPP_QUERY_PROBE (phase stamps in the search
# kernel); every other object is the shipped one (pytorch_points_amd/build/*.o, made by _build.build()).
set -e
cd "$(dirname "$0")/.."
python -c "from pytorch_points_amd import _build; _build.build()"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fPIC -std=c++17 -Wall -Wno-unused-function \
  -DPP_QUERY_PROBE $PP_PROBE_FLAGS -Iinclude -Ipytorch_points_amd/csrc -c pytorch_points_amd/csrc/chamfer_grid.hip -o /tmp/chamfer_grid_probe.o
objs=$(ls pytorch_points_amd/build/*.o | grep -v chamfer_grid.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/chamfer_grid_probe.o -o ${PP_PROBE_OUT:-tools/libpp_hip_probe.so}
