#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/spx_trace
rm -rf $OUT && mkdir -p $OUT
( while true; do sleep 50; echo "[trace] running" >> $OUT/progress.log; done ) &
HB=$!
timeout -k 10 700 rocprofv3 --kernel-trace --stats -d $OUT/raw --output-format csv -- python3 $R/tools/spx_c2.py > $OUT/run.json 2> $OUT/stderr.log
kill $HB
find $OUT/raw -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_spx_c2.csv \;
rm -rf $OUT/raw
head -12 $OUT/kernel_stats_spx_c2.csv | cut -c1-60,150-260
cat $OUT/run.json | cut -c1-200
