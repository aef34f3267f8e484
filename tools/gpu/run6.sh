python -m pytest tests/test_gpu_bandlu.py tests/test_gpu_crossover_band.py -x -q > gpurun_out/r7_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r7_pytest.log
tail -4 gpurun_out/r7_pytest.log
SX_SPX_TRACE=1 timeout -k 10 120 python tools/lp_e2e.py n1 > gpurun_out/r7_n1.json 2> gpurun_out/r7_n1_trace.txt; echo "n1 rc=$?"
cat gpurun_out/r7_n1.json; grep -v "^\[sx_crossover_band\]   round" gpurun_out/r7_n1_trace.txt | tail -12
SX_SPX_TRACE=1 timeout -k 10 600 python tools/lp_e2e.py n1 m=1000000 n=10000000 > gpurun_out/r7_c5.json 2> gpurun_out/r7_c5_trace.txt; echo "c5 rc=$?"
cat gpurun_out/r7_c5.json
grep -v "round [0-9]*: status 0" gpurun_out/r7_c5_trace.txt | tail -30
