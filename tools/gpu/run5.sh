python -m pytest tests/test_gpu_bandlu.py tests/test_gpu_crossover_band.py -x -q > gpurun_out/r5_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_pytest.log
tail -4 gpurun_out/r5_pytest.log
bash tools/prof_lp.sh c5 n1 m=1000000 n=10000000 > gpurun_out/r5_prof_c5.txt 2>&1; tail -30 gpurun_out/r5_prof_c5.txt | cut -c1-200
