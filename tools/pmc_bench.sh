#!/bin/bash
# PMC passes over bench.py (separate runs per counter group, as MI355X_MICROARCH.md prescribes); summaries under
# gpurun_out/pmc_bench/ -- copy what is to be judged into profiles/rNN/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_bench
rm -rf $OUT && mkdir -p $OUT
ARGS="$R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-uniform --no-crossover $EXTRA"
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/req --output-format csv -- python3 $ARGS > $OUT/req.log 2>&1
HASH=$(cd $R && python3 -c "import bench; print(bench.source_hash())")
python3 $R/tools/pmc_summary.py --json $OUT/kernel_traffic.json --workload c5/staircase --source-hash $HASH \
    --command "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum> -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-uniform (three passes)" \
    $OUT/fetch $OUT/write $OUT/req > $OUT/summary.txt 2>&1
rm -rf $OUT/fetch $OUT/write $OUT/req
cat $OUT/summary.txt
