#!/bin/bash
# tools/refresh_profiles.sh  -- on the GPU box: bench lines, rocprofv3 kernel stats and PMC passes
# for the three BASELINE workloads, into gpurun_out/refresh/ (copy what is to be kept into profiles/).
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/refresh
rm -rf $O; mkdir -p $O
timeout 400 python3 bench.py > $O/bench_chamfer_n1.json 2> $O/bench_chamfer.err
timeout 300 python3 bench.py --workload fps --steps 5 --warmup 2 > $O/bench_fps.json 2> $O/bench_fps.err
timeout 300 python3 bench.py --workload ball_group --with-backward --steps 20 --warmup 5 > $O/bench_ball_group.json 2> $O/bench_bg.err
for w in chamfer fps ball_group; do
  extra=""; [ $w = chamfer ] && extra="--launch eager --no-cpu-baseline --no-extras"; [ $w = ball_group ] && extra="--with-backward"
  (timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 bench.py --workload $w --steps 10 --warmup 3 $extra > $O/prof_$w.log 2>&1)
  f=$(find $O/prof_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${w}_kernel_stats.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  (timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/ch_$c -- python3 bench.py --steps 3 --warmup 1 --launch eager --no-cpu-baseline --no-extras > $O/pmc_ch_$c.log 2>&1)
  (timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/bg_$c -- python3 bench.py --workload ball_group --with-backward --steps 3 --warmup 1 > $O/pmc_bg_$c.log 2>&1)
done
python3 tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
rm -rf $O/prof_chamfer $O/prof_fps $O/prof_ball_group
find $O/pmc -name '*agent_info.csv' -delete
ls -la $O
