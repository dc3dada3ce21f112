#!/bin/bash
# phase probe + SQ counters of the search kernel, wave-private form (-1) against the tile form (512)
python tools/query_probe.py -1 512 256 > gpurun_out/qprobe1.log 2>&1
C1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS"
C2="SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
bash tools/pmc_multi.sh w "$C1" "$C2" -- tools/fwd_loop.py -1 sphere 30 > gpurun_out/pmc_wave.log 2>&1
bash tools/pmc_multi.sh t "$C1" "$C2" -- tools/fwd_loop.py 512 sphere 30 > gpurun_out/pmc_tile.log 2>&1
tail -3 gpurun_out/pmc_tile.log
