SX_SPX_TRACE=1 timeout -k 10 400 python tools/k1_netlib_bench.py 2>&1 | grep -E "sx_window|swizzle=1 window=-1|swizzle=1 window= 4" | head -6
python -m pytest tests/test_gpu_lp_parity.py tests/test_gpu_property.py tests/test_gpu_slabs.py tests/test_gpu_lp_api.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -3
SX_SPX_TRACE=1 python bench.py --no-crossover --no-cpu-baseline --no-uniform --steps 10 2>&1 | grep -E "sx_window|\"metric\"" | cut -c1-900
