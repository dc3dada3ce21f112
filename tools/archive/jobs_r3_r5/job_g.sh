#!/bin/bash
# one rank over RCCL (PP_BENCH_FORCE_DIST=1): direct ncclAllGather, c10d from C++, Python -- all issued by the calling thread
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 PP_BENCH_FORCE_DIST=1
for mode in rccl native python; do
  PP_SHARD_EXCHANGE=$mode timeout 600 python bench.py --gpus 1 --steps 300 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/bench_dist_$mode.json 2> gpurun_out/bench_dist_$mode.err
done
python - <<'PY'
import json
for n in ("rccl", "native", "python"):
    try:
        d=json.loads(open("gpurun_out/bench_dist_%s.json" % n).read().strip().splitlines()[-1])
        print(n, "ms_per_step", round(d["ms_per_step"],4), "compute_ms", round(d.get("compute_ms"),4), "exchange_ms", round(d.get("exchange_ms"),4), "exchange_gpu_us", round(d.get("exchange_gpu_us") or 0, 1), "issue_us", d.get("exchange_issue_us"), "|", d.get("exchange_issue"))
    except Exception as e:
        print(n, "failed", e); print(open("gpurun_out/bench_dist_%s.err" % n).read()[-1500:])
PY
timeout 900 python -m pytest tests/test_gpu_shard.py -m gpu -x -q 2>&1 | tail -15
