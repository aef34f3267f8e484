#!/bin/bash
# rocprofv3 --kernel-trace --stats over one LP crossover (tools/lp_e2e.py): per-kernel totals of the first-order
# stage, the band LU and the tableau simplex.  usage (on the GPU box): tools/prof_lp.sh NAME lp_e2e-args...
set -e
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_lp
mkdir -p $OUT
rm -rf $OUT/raw_$NAME
rocprofv3 --kernel-trace --stats -d $OUT/raw_$NAME --output-format csv -- python3 $R/tools/lp_e2e.py "$@" > $OUT/$NAME.json 2> $OUT/$NAME.stderr
find $OUT/raw_$NAME -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$NAME.csv \;
rm -rf $OUT/raw_$NAME
head -25 $OUT/kernel_stats_$NAME.csv
tail -2 $OUT/$NAME.json | cut -c1-400
