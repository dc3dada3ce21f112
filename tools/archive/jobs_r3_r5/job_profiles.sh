#!/bin/bash
# round 3 profile set: bench lines, kernel stats, HBM traffic counters (separate passes), SQ counters of the forward's
# kernels, phase probes -- into gpurun_out/refresh (copied to profiles/r3 by hand)
bash tools/refresh_profiles.sh > gpurun_out/refresh_ls.txt 2>&1
C1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS"
C2="SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
bash tools/pmc_multi.sh r3 "$C1" "$C2" -- tools/fwd_loop.py 0 sphere 30 > gpurun_out/refresh/chamfer_sq_counters.txt 2>&1
PP_PROBE_LIB=libpp_hip_probe_a.so python tools/query_probe.py 0 > gpurun_out/refresh/stage_a_phase_probe.txt 2>&1
python tools/tile_modes.py > gpurun_out/refresh/search_forms.txt 2>&1
