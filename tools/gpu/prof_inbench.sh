#!/bin/bash
# Item "in-bench slowdown": kernel-level evidence.  The headline LP crossover (3 calls) inside bench.py's process state
# (scoring loop, CPU baseline, uniform workload done before it) against the same 3 calls in a process of their own, both
# under rocprofv3 --kernel-trace --stats with direct launches (SX_NO_GRAPH): per-kernel average durations side by side.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_inbench
rm -rf $OUT && mkdir -p $OUT
export SX_NO_GRAPH=1
SX_BENCH_ONLY_LP_1E6=1 rocprofv3 --kernel-trace --stats -d $OUT/raw_bench --output-format csv -- python3 $R/bench.py --steps 5 > $OUT/bench.json 2> $OUT/bench.err
find $OUT/raw_bench -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_inbench.csv \;
rm -rf $OUT/raw_bench
rocprofv3 --kernel-trace --stats -d $OUT/raw_alone --output-format csv -- python3 $R/tools/lp_e2e.py n1 gpp_reps=3 > $OUT/alone.json 2> $OUT/alone.err
find $OUT/raw_alone -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_alone.csv \;
rm -rf $OUT/raw_alone
tail -c 600 $OUT/bench.json; echo; cat $OUT/alone.json
