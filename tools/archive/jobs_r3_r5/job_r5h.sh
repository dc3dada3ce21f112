#!/bin/bash
# round 5, VERDICT r4 #8: what the group_points forward's remaining 12 % to the store ceiling is: write requests by size,
# write stalls, and the same counters for the store-ceiling kernel (pp_debug_store_ceiling) in the same run
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r5h; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_WR[A-Z0-9_]*\|TCC_EA0_WRREQ[A-Za-z0-9_]*\|TCC_[A-Z0-9_]*STALL[A-Z0-9_]*\|TCC_WRITE[A-Za-z0-9_]*\|TCP_[A-Z_]*WRITE[A-Z_]*" | sort -u | head -60 > $O/counters_available.txt
cat $O/counters_available.txt | tr '\n' ' '; echo
bash tools/pmc_multi.sh gp "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum" "TCC_EA0_WR_UNCACHED_32B_sum TCC_WRITEBACK_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM_WR SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES" -- tools/gp_probe.py 0 2>&1 | grep -v amdgpu.ids > $O/gp_counters.txt
cat $O/gp_counters.txt | head -60
