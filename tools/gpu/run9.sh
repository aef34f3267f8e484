python -m pytest tests/test_gpu_bandlu.py tests/test_gpu_crossover_band.py tests/test_gpu_denselu.py -x -q -s > gpurun_out/r9_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r9_pytest.log
grep "hostile structure" gpurun_out/r9_pytest.log; tail -4 gpurun_out/r9_pytest.log
timeout -k 10 120 python tools/lp_e2e.py n1 gpp_reps=2 > gpurun_out/r9_n1.json 2> /dev/null; cat gpurun_out/r9_n1.json
SX_SPX_TRACE=1 timeout -k 10 600 python tools/lp_e2e.py n1 m=1000000 n=10000000 > gpurun_out/r9_c5.json 2> gpurun_out/r9_c5_trace.txt; echo "c5 rc=$?"
cat gpurun_out/r9_c5.json
grep -v "round [0-9]*: status 0" gpurun_out/r9_c5_trace.txt | grep -v "round [0-9]: \(duals\|reduced\)" | tail -30
