#!/bin/bash
# everything the driver runs at round end: the GPU suite, smoke(), the default bench line
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_final.log 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_final.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -3 gpurun_out/pytest_final.log; tail -2 gpurun_out/smoke_final.log; python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "median", d.get("ms_per_step_events_median"), "fwd", d.get("fwd_ms"), "frac", d["roofline"]["frac"])
print(d.get("other_distributions_fwd_ms")); print(d["fps"]["us_per_pick"], d["ball_group"]["ball_query_ms"], d["ball_group"]["group_points_ms"])
PY
