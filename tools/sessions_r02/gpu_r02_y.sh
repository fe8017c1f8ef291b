#!/bin/bash
# Round 2, GPU session Y: final measurements on the final tree: rocprofv3 kernel stats of a short bench, the driver's
# bench line (20 steps, 5 warm-up), the default bench invocation (wall time).
export TMPDIR=/tmp
O=gpurun_out/r02y
mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r02y -o r02 -- python /root/repo/bench.py --steps 6 --warmup 1 --no_cpu_baseline --no_kernels > /root/repo/$O/bench_under_profiler.log 2>&1
find /tmp/prof_r02y -name "*kernel_stats.csv" -exec cp {} /root/repo/$O/kernel_stats_bench_steps6.csv \;
cd /root/repo
head -14 $O/kernel_stats_bench_steps6.csv | cut -c1-150
echo "== bench, the driver's arguments (20 steps, 5 warm-up)"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json.log 2> $O/bench_steps20.err; tail -c 600 $O/bench_steps20.json.log; tail -2 $O/bench_steps20.err
echo "== bench, no flags"
/usr/bin/time -v timeout 900 python bench.py > $O/bench_default.json.log 2> $O/bench_default.err; grep "Elapsed" $O/bench_default.err; head -c 300 $O/bench_default.json.log
