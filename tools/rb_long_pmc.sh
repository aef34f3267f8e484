#!/bin/bash
# PMC passes over tools/rb_bench.py: HBM-side reads / writes of the long-row pre-pass and the walks
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/rbl_pmc
rm -rf $OUT && mkdir -p $OUT
ARGS="$R/tools/rb_bench.py --rounds 1 --reps 3"
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/req --output-format csv -- python3 $ARGS > $OUT/req.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/req > $OUT/summary.txt 2>&1
rm -rf $OUT/fetch $OUT/write $OUT/req
grep -E "rb_score_rows|long_products|k_score_rows" $OUT/summary.txt
