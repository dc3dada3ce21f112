#!/bin/bash
# chamfer_slab.hip: kernel durations (rocprofv3 --kernel-trace --stats) of the variants that leave after phase n
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for n in ${VARIANTS:-1 2 3 4 5 6 full}; do
  if [ $n = full ]; then unset PP_LIB; else export PP_LIB=tools/libpp_hip_slab$n.so; [ -f $PP_LIB ] || continue; fi
  rm -rf gpurun_out/slabprof_$n
  PP_TILE_MODES=-2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/slabprof_$n -- python3 tools/tile_modes.py sphere > /dev/null 2>&1
  f=$(find gpurun_out/slabprof_$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] || { echo "variant $n: no stats"; continue; }
  python3 - "$f" "$n" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Name"]
    if "chamfer_slab" in nm or (sys.argv[2] == "full" and ("grid_build" in nm or "grid_query_wave" in nm)):
        print("variant %s: %-48s calls %s avg_ns %s min %s max %s" % (sys.argv[2], nm[:48], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
PY
  rm -rf gpurun_out/slabprof_$n
done
