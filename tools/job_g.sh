#!/bin/bash
# one rank over RCCL (PP_BENCH_FORCE_DIST=1): the native exchange and the Python one
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 PP_BENCH_FORCE_DIST=1
timeout 600 python bench.py --gpus 1 --steps 300 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/bench_dist_native.json 2> gpurun_out/bench_dist_native.err
PP_SHARD_EXCHANGE=python timeout 600 python bench.py --gpus 1 --steps 300 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/bench_dist_python.json 2> gpurun_out/bench_dist_python.err
python - <<'PY'
import json
for n in ("native", "python"):
    try:
        d=json.loads(open("gpurun_out/bench_dist_%s.json" % n).read().strip().splitlines()[-1])
        print(n, "ms_per_step", round(d["ms_per_step"],4), "compute_ms", d.get("compute_ms"), "exchange_ms", d.get("exchange_ms"), d["config"].get("parallelism"))
    except Exception as e:
        print(n, "failed", e); print(open("gpurun_out/bench_dist_%s.err" % n).read()[-800:])
PY
