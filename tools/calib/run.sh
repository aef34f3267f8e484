#!/bin/bash
# FETCH_SIZE calibration on the GPU box; output: gpurun_out/calib/summary.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/calib
rm -rf $OUT && mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -o $OUT/fetch_calib $R/tools/calib/fetch_calib.hip
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
grep -o "TCC_EA0_RD[A-Z0-9_]*\|TCC_REQ[A-Za-z0-9_]*\|TCC_READ[A-Za-z0-9_]*" $OUT/counters.txt | sort -u > $OUT/tcc_counters.txt || true
rm -f $OUT/counters.txt
rocprofv3 --pmc FETCH_SIZE -d $OUT/a --output-format csv -- $OUT/fetch_calib > $OUT/a.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/b --output-format csv -- $OUT/fetch_calib > $OUT/b.log 2>&1 || true
python3 $R/tools/pmc_summary.py $OUT/a $OUT/b > $OUT/summary.txt 2>&1
rm -rf $OUT/a $OUT/b $OUT/fetch_calib
cat $OUT/summary.txt $OUT/tcc_counters.txt
