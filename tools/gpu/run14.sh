timeout -k 10 300 python tools/rb_long_xcd_bench.py --workload shard 2>&1 | tail -4
timeout -k 10 300 python tools/rb_long_xcd_bench.py --workload netlib 2>&1 | tail -4
python -m pytest tests/test_gpu_rowblock.py tests/test_gpu_cg.py tests/test_gpu_lp_parity.py -x -q 2>&1 | tail -3
timeout -k 10 600 python tools/lp_e2e.py n1 m=1000000 n=10000000 2>/dev/null | cut -c1-330
