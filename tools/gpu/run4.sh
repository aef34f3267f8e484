python -m pytest tests/test_gpu_bandlu.py tests/test_gpu_crossover_band.py -x -q > gpurun_out/r4_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_pytest.log
tail -4 gpurun_out/r4_pytest.log
for blocks in 1 0; do
  if [ $blocks = 1 ]; then export SX_BAND_BLOCKS=1; else unset SX_BAND_BLOCKS; fi
  SX_SPX_TRACE=1 timeout -k 10 120 python tools/lp_e2e.py n1 > gpurun_out/r4_n1_$blocks.json 2> gpurun_out/r4_n1_trace_$blocks.txt; echo "n1 rc=$?"
  cat gpurun_out/r4_n1_$blocks.json; grep -v "round [0-9]*: status 0" gpurun_out/r4_n1_trace_$blocks.txt | grep -v "^\[sx_crossover_band\]   round" | tail -14
done
SX_SPX_TRACE=1 timeout -k 10 600 python tools/lp_e2e.py n1 m=1000000 n=10000000 > gpurun_out/r4_c5.json 2> gpurun_out/r4_c5_trace.txt; echo "c5 rc=$?"
cat gpurun_out/r4_c5.json
grep -v "round [0-9]*: status 0" gpurun_out/r4_c5_trace.txt | tail -32
