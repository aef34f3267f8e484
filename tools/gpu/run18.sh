SX_SPX_TRACE=1 timeout -k 10 400 python tools/k1_netlib_bench.py 2>&1 | grep -E "sx_window|swizzle=1 window=-1 chunk=4096" | head -4
python -m pytest tests -m gpu -x -q > gpurun_out/r18_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r18_pytest.log
grep "hostile structure" gpurun_out/r18_pytest.log; tail -4 gpurun_out/r18_pytest.log
