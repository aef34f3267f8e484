#!/bin/bash
# rocprofv3 --kernel-trace --stats over bench.py (the walks only: the whole-crossover legs launch millions of small
# simplex kernels and are timed by bench.py itself); keeps the per-kernel statistics, drops the raw trace.
# Output under gpurun_out/trace_bench/ -- copy what is to be judged into profiles/rNN/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_bench
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/raw --output-format csv -- python3 $R/bench.py --no-crossover --no-cpu-baseline $EXTRA > $OUT/bench_line_under_rocprof.json 2> $OUT/stderr.log
find $OUT/raw -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_bench_c5.csv \;
rm -rf $OUT/raw
head -12 $OUT/kernel_stats_bench_c5.csv
