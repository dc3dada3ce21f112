#!/bin/bash
# tools/job.sh <what> -- the gpurun jobs of this repository, one script (run on the GPU box from the root of the tree:
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/job.sh final').  Output under gpurun_out/.
#   final     everything the driver runs at round end: the GPU suite, smoke(), the default bench line
#   tests     the GPU suite only          fuzz      the long fuzz run (PP_FUZZ_SEEDS=1500)
#   fps       FPS: parity tests, the round probe, timings over shapes / clouds, the config-3 bench line
#   fpsquick  the same, the bucketed kernel at config 3 only
#   shard     the batch-sharded exchange: one-rank RCCL paths, the distributed bench line's logic
#   exchange  the exchange forms on a one-rank RCCL group, 300 steps each (profiles/r<N>/exchange_one_rank_runs.txt)
#   benchline the default bench line once more (the eager number follows the host)
#   chamfer   the grid search's parity tests, then the forward on the other clouds (tools/far_time.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
case "$1" in
final)
  timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_final.log 2>&1
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_final.log 2>&1
  timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
  tail -3 gpurun_out/pytest_final.log; tail -2 gpurun_out/smoke_final.log
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "median", d.get("ms_per_step_events_median"), "fwd", d.get("fwd_ms"), "frac", d["roofline"]["frac"])
print(d.get("other_distributions_fwd_ms"))
print(d["fps"]["ms_per_step"], d["fps"]["us_per_pick"], d["ball_group"]["ball_query_ms"], d["ball_group"]["group_points_ms"])
PY
  ;;
tests)
  timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/pytest_all.txt ;;
fuzz)
  PP_FUZZ_SEEDS=1500 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/pytest_fuzz.log 2>&1
  tail -5 gpurun_out/pytest_fuzz.log ;;
fps|fpsquick)
  timeout 1500 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_golden.py tests/test_gpu_nonfinite.py tests/test_gpu_fuzz.py -m gpu -x -q -k "fps or furthest or FPS" > gpurun_out/pytest_fps.log 2>&1
  tail -5 gpurun_out/pytest_fps.log
  (timeout 120 ./tools/fps_bucket_probe 16 65536 4096; [ "$1" = fps ] && PP_PROBE_CHAIN=1 timeout 120 ./tools/fps_bucket_probe 16 65536 4096) > gpurun_out/fps_bucket_probe.txt 2>&1
  grep -v "^ wave  *[4-9]\|^ wave 1[0-2]" gpurun_out/fps_bucket_probe.txt
  timeout 600 python tools/fps_time.py $([ "$1" = fpsquick ] && echo quick) > gpurun_out/fps_time.txt 2>&1
  cat gpurun_out/fps_time.txt
  if [ "$1" = fps ]; then
    timeout 300 python bench.py --workload fps --steps 5 --warmup 2 > gpurun_out/bench_fps.json 2> gpurun_out/bench_fps.err
    cat gpurun_out/bench_fps.json
  fi ;;
shard)
  timeout 1500 python -m pytest tests/test_gpu_shard.py tests/test_gpu_bench_contract.py -m gpu -x -q > gpurun_out/pytest_shard.log 2>&1
  tail -25 gpurun_out/pytest_shard.log ;;
exchange)
  export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 PP_BENCH_FORCE_DIST=1
  for mode in native p2p rccl rccl_p2p python; do
    PP_SHARD_EXCHANGE=$mode timeout 600 python bench.py --gpus 1 --steps 300 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/bench_dist_$mode.json 2> gpurun_out/bench_dist_$mode.err
  done
  python - <<'PY'
import json
for n in ("native", "p2p", "rccl", "rccl_p2p", "python"):
    try:
        d = json.loads(open("gpurun_out/bench_dist_%s.json" % n).read().strip().splitlines()[-1])
        print(n, "ms_per_step", round(d["ms_per_step"], 4), "compute_ms", round(d.get("compute_ms"), 4), "exchange_ms", round(d.get("exchange_ms"), 4),
              "exchange_gpu_us", round(d.get("exchange_gpu_us") or 0, 1), "issue_us", d.get("exchange_issue_us"), "|", d.get("exchange_issue"))
    except Exception as e:
        print(n, "failed", e); print(open("gpurun_out/bench_dist_%s.err" % n).read()[-1500:])
PY
  ;;
benchline)
  mkdir -p gpurun_out/benchline
  timeout 600 python3 bench.py > gpurun_out/benchline/bench_chamfer_n1.json 2> gpurun_out/benchline/err.txt
  python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/benchline/bench_chamfer_n1.json").read().strip().splitlines()[-1])
print("eager %.4f ext %.4f host_bound %s" % (d["ms_per_step"], d["launch_modes_ms_per_step"]["ext"], d.get("host_bound")))
PY
  ;;
chamfer)
  timeout 2400 python -m pytest tests/test_gpu_chamfer_grid.py tests/test_gpu_chamfer.py -m gpu -x -q > gpurun_out/pytest_chamfer.log 2>&1
  tail -5 gpurun_out/pytest_chamfer.log
  timeout 600 python tools/far_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/far_time.txt ;;
*) echo "usage: tools/job.sh final|tests|fuzz|fps|fpsquick|shard|exchange|benchline|chamfer"; exit 2 ;;
esac
