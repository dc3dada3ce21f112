#!/bin/bash
# profiles/r5/near_field_stages_ab.txt: the forward of nndistance on the clouds of tools/tile_modes.py with each of round 5's
# second-half changes switched off (variant libraries of tools/build_variant_lib.sh, built from commit 3a4d3cc), one box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r5ab
(
echo "# commit 3a4d3cc -- tools/job_r5ab.sh: PP_LIB=<variant> PP_TILE_MODES=512 python3 tools/tile_modes.py (forward, ms; 'ok' = bit-identical to the every-pair kernel)"
for v in "" v_noball v_laneball v_nocubepool v_noafter v_noonerow v_farlate v_oldbuild; do
  case "$v" in
    "") echo "== as shipped";;
    v_noball) echo "== without any of it: -DPP_LANE_BALL=0 -DPP_POOLED_CUBE=0 -DPP_SCAN_ONE_ROW=0 -DPP_MEMBER_CUT_MIN=8 -DPP_SERIAL_FAR=65 -DPP_BUILD_BALANCE=0 -DPP_BUILD_SPILL=0 -DPP_BUILD_JUMP=0 (the DPP / v_sqrt changes of the far-field stages stay)";;
    v_laneball) echo "== the balls a lane per query instead of pooled: -DPP_POOLED_BALL=0 (the cubes then a lane per query too)";;
    v_nocubepool) echo "== the cubes of radius 1 and 2 a lane per query: -DPP_POOLED_CUBE=0";;
    v_noafter) echo "== no second ball stage behind the cubes of radius 1: -DPP_BALL_AFTER_CUBE=0";;
    v_noonerow) echo "== the whole-wave scans search every candidate's row: -DPP_SCAN_ONE_ROW=0";;
    v_farlate) echo "== far pending lanes to the group search from 24 on, member cut from 8 candidates per member: -DPP_SERIAL_FAR=65 -DPP_MEMBER_CUT_MIN=8";;
    v_oldbuild) echo "== the build without histogram-dealt slabs, register scatter, occupancy-sized steps: -DPP_BUILD_BALANCE=0 -DPP_BUILD_SPILL=0 -DPP_BUILD_JUMP=0";;
  esac
  if [ -n "$v" ]; then export PP_LIB=tools/libpp_hip_$v.so; else unset PP_LIB; fi
  PP_TILE_MODES=512 timeout 600 python3 tools/tile_modes.py sphere cube gaussian shapenet_like two_scales blobs8 disjoint plane line 2>&1 | grep -v amdgpu.ids
done
unset PP_LIB
echo "== labeled Chamfer (tools/labeled_time.py), as shipped / the lane forms (-DPP_POOLED_BALL=0)"
timeout 600 python3 tools/labeled_time.py 2>&1 | grep -v amdgpu.ids
PP_LIB=tools/libpp_hip_v_laneball.so timeout 600 python3 tools/labeled_time.py 2>&1 | grep -v amdgpu.ids
) > gpurun_out/r5ab/near_field_stages_ab.txt 2>&1
tail -5 gpurun_out/r5ab/near_field_stages_ab.txt
