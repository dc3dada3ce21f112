#!/bin/bash
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest20.log 2>&1
timeout 600 python bench.py > gpurun_out/bench20.json 2> gpurun_out/bench20.err
tail -5 gpurun_out/pytest20.log; python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench20.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "median", d.get("ms_per_step_events_median"), "fwd", d.get("fwd_ms"))
print(d.get("other_distributions_fwd_ms")); print(d["roofline"]["frac"], d["roofline"].get("traffic"))
PY
