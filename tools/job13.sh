#!/bin/bash
# item 4: the exchange as one native call -- one-rank RCCL group, step with and without the exchange
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 PP_BENCH_FORCE_DIST=1
python bench.py --steps 300 --no-extras --no-cpu-baseline > gpurun_out/bench_dist_native.json 2> gpurun_out/bench_dist_native.err
MASTER_PORT=29573 PP_SHARD_EXCHANGE=python python bench.py --steps 300 --no-extras --no-cpu-baseline > gpurun_out/bench_dist_python.json 2> gpurun_out/bench_dist_python.err
for f in native python; do python3 -c "
import json
d=json.loads(open('gpurun_out/bench_dist_$f.json').read().strip().split('\n')[-1])
print('$f', 'ms_per_step', round(d['ms_per_step'],4), 'compute_ms', d.get('compute_ms'), 'exchange_ms', d.get('exchange_ms'))
"; done | tee gpurun_out/bench_dist_summary.txt
tail -3 gpurun_out/bench_dist_native.err
