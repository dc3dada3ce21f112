#!/bin/bash
# tools/prof.sh <tag> <python script and args...>  -- rocprofv3 kernel stats (csv) into gpurun_out/prof_<tag>/
# (always csv: the default .db output with --stats does not finish on this pool)
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 "$@" > gpurun_out/prof_$tag.log 2>&1
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
test -n "$f" && head -${PROF_LINES:-10} "$f" | cut -d, -f1-8
