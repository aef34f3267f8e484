#!/bin/bash
# SQ counters of K2 over the column-blocked layout at config 5 (lp_shard): where do the waves' cycles go?
R=$(pwd); OUT=$R/gpurun_out/k2_sq; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
pick() { for c in "$@"; do grep -qw "$c" $OUT/avail.txt && echo -n "$c "; done; }
P1=$(pick SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU)
P2=$(pick SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU)
P2=$(echo $P2 | cut -d' ' -f1-8)
echo "pass 1: $P1"; echo "pass 2: $P2"
rocprofv3 --pmc $P1 -d $OUT/p1 --output-format csv -- python3 $R/tools/rb_long_rows_bench.py --workload shard --option rb_long_rows 64 > $OUT/p1.log 2>&1
rocprofv3 --pmc $P2 -d $OUT/p2 --output-format csv -- python3 $R/tools/rb_long_rows_bench.py --workload shard --option rb_long_rows 64 > $OUT/p2.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rb_score_rows" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in sorted(acc.items()):
        print(f"{p} k_rb_score_rows {k:28s} n={n:4d} mean={s / max(n, 1):.4e}")
PY
rm -rf $OUT/p1 $OUT/p2
