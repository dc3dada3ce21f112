#!/bin/bash
# the default bench line once more on another box (the eager number follows the host: README) -> gpurun_out/benchline/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out/benchline
timeout 600 python3 bench.py > gpurun_out/benchline/bench_chamfer_n1.json 2> gpurun_out/benchline/err.txt
python3 - <<'PY'
import json
p = "gpurun_out/benchline/bench_chamfer_n1.json"
d = json.loads(open(p).read().strip().splitlines()[-1])
d["_commit"] = "2cfc68e"
d["_generated_by"] = "tools/job_benchline.sh"
open(p, "w").write(json.dumps(d) + "\n")
print("eager %.4f ext %.4f host_bound %s" % (d["ms_per_step"], d["launch_modes_ms_per_step"]["ext"], d.get("host_bound")))
PY
