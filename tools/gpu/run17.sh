SX_SPX_TRACE=1 timeout -k 10 400 python tools/k1_netlib_bench.py 2>&1 | grep -E "sx_window|swizzle=1 window=-1 chunk=4096|swizzle=1 window= 4 chunk=4096|swizzle=1 window= 0 chunk=4096|swizzle=0 window=-1 chunk=4096" | head -6
python -m pytest tests/test_gpu_lp_parity.py tests/test_gpu_property.py tests/test_gpu_slabs.py tests/test_gpu_rowblock.py tests/test_gpu_cg.py -x -q 2>&1 | tail -3
