set -x
python -m pytest tests/test_gpu_crossover_band.py tests/test_gpu_bandlu.py tests/test_gpu_denselu.py -x -q > gpurun_out/r1_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r1_pytest.log
tail -5 gpurun_out/r1_pytest.log
SX_SPX_TRACE=1 timeout -k 10 300 python tools/lp_e2e.py n1 gpp_reps=2 > gpurun_out/r1_n1.json 2> gpurun_out/r1_n1_trace.txt; echo "n1 rc=$?"
cat gpurun_out/r1_n1.json
timeout -k 10 200 python tools/denselu_bench.py 1000 4000 10000 > gpurun_out/r1_denselu.txt 2>&1; cat gpurun_out/r1_denselu.txt
