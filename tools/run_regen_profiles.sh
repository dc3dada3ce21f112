#!/bin/bash
# tools/run_regen_profiles.sh -- in the build container: build everything at HEAD, run tools/regen_profiles_r4.sh on a GPU
# box through gpurun, and copy what it produced into profiles/r4/ (one script regenerates every measured
# file of the round from the final commit, the commit named in each).
set -e
cd "$(dirname "$0")/.."
if [ -n "$(git status --porcelain -- pytorch_points_amd include bench.py tools oracle)" ]; then echo "commit first: the profiles name a commit"; exit 1; fi
C=$(git rev-parse --short HEAD)
python -c "import __graft_entry__ as g; g.build()"
./tools/build_fps_bucket_probes.sh > /dev/null 2>&1 || true
rm -f tools/libpp_hip_*.so
for n in 1 2 3 4 5 6; do SRC=chamfer_slab bash tools/build_variant_lib.sh slab$n -DPP_SLAB_STOP=$n; done
/usr/local/graft/bin/gpurun --timeout 3000 -- "PP_COMMIT=$C bash tools/regen_profiles_r4.sh"
mkdir -p profiles/r4
cp gpurun_out/r4/* profiles/r4/
echo "profiles/r4 refreshed from commit $C"
