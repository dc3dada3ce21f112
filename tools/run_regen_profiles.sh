#!/bin/bash
# tools/run_regen_profiles.sh -- in the build container: build everything at HEAD, run tools/regen_profiles.sh on a GPU
# box through gpurun, and copy what it produced into profiles/r<round>/ (round = $1, default 6; VERDICT r3 #7: one script regenerates every measured
# file of the round from the final commit, the commit named in each).
set -e
cd "$(dirname "$0")/.."
if [ -n "$(git status --porcelain -- pytorch_points_amd include bench.py tools oracle)" ]; then echo "commit first: the profiles name a commit"; exit 1; fi
C=$(git rev-parse --short HEAD)
R=${1:-6}
python -c "import __graft_entry__ as g; g.build()"
./tools/build_fps_bucket_probes.sh > /dev/null 2>&1 || true
rm -f tools/libpp_hip_*.so
PP_PROBE_FLAGS=-DPP_QUERY_PROBE_NO_GROUP_STATS PP_PROBE_OUT=tools/libpp_hip_probe_light.so bash tools/build_probe_lib.sh > /dev/null 2>&1 || true
/usr/local/graft/bin/gpurun --timeout 3000 -- "PP_COMMIT=$C PP_ROUND=$R bash tools/regen_profiles.sh"
mkdir -p profiles/r$R
cp gpurun_out/r$R/* profiles/r$R/
echo "profiles/r$R refreshed from commit $C"
