timeout -k 10 400 python tools/rb_long_rows_bench.py --workload shard --option rb_dense_min 512 256 1024 2048 1000000 2>&1 | tail -11
