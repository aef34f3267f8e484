#!/bin/bash
# PMC passes over tools/rb_bench.py (one config per run keeps the CSVs small); results under gpurun_out/rbprof/
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/rbprof
mkdir -p $OUT
CFG=${1:-512:4096:4096:256:2}
ARGS="$R/tools/rb_bench.py --rounds 1 --reps 3 --configs $CFG"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/a --output-format csv -- python3 $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU -d $OUT/b --output-format csv -- python3 $ARGS > $OUT/b.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $OUT/c --output-format csv -- python3 $ARGS > $OUT/c.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_LDS -d $OUT/d --output-format csv -- python3 $ARGS > $OUT/d.log 2>&1 || true
python3 $R/tools/pmc_summary.py $OUT/a $OUT/b $OUT/c $OUT/d > $OUT/summary_${CFG//:/_}.txt 2>&1
grep -E "rb_walk|score_rows" $OUT/summary_${CFG//:/_}.txt
