# pivot tolerance sweep of the guessed basis at the headline size (trace lines that matter)
for cfg in "1e-3 1e-2" "1e-3 1e-1" "1e-2 1e-1" "1e-7 1e-2" "1e-1 3e-1" "3e-1 5e-1"; do
  set -- $cfg
  echo "=== tolB=$1 tolS=$2" >> gpurun_out/r2_sweep.txt
  SX_BAND_PIVTOL_B=$1 SX_BAND_PIVTOL_S=$2 SX_SPX_TRACE=1 timeout -k 10 60 python tools/lp_e2e.py n1 > gpurun_out/r2_tmp.json 2> gpurun_out/r2_tmp_trace.txt
  cat gpurun_out/r2_tmp.json >> gpurun_out/r2_sweep.txt
  grep -E "band LU done|Schur complement factored|basic solution deviates|outside their bounds:|tracked columns \(tableau|round .*status|done:" gpurun_out/r2_tmp_trace.txt >> gpurun_out/r2_sweep.txt
done
python -m pytest tests/test_gpu_crossover_band.py tests/test_gpu_bandlu.py -x -q > gpurun_out/r2_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest.log
tail -3 gpurun_out/r2_pytest.log
