#!/bin/bash
# tools/fps_bucket_probe (phase marks) and tools/fps_bucket_probe_d<bits> (no marks; one link of the chain doubled)
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -std=c++17 -Iinclude -Ipytorch_points_amd/csrc"
/opt/rocm/bin/hipcc $F tools/fps_bucket_probe.hip -o tools/fps_bucket_probe 2>/dev/null &
for d in 0 1 2 4 8 16 32; do
  /opt/rocm/bin/hipcc $F -DPP_FPSB_NOMARKS -DPP_FPSB_DOUBLE=$d tools/fps_bucket_probe.hip -o tools/fps_bucket_probe_d$d 2>/dev/null &
done
for st in 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc $F -DPP_FPSB_NOMARKS -DPP_FPSB_STOP=$st tools/fps_bucket_probe.hip -o tools/fps_bucket_probe_s$st 2>/dev/null &
done
wait
ls -la tools/fps_bucket_probe*
