python -m pytest tests/test_gpu_crossover_band.py tests/test_gpu_rowblock.py -x -q 2>&1 | tail -2
timeout -k 10 120 python tools/lp_e2e.py n1 gpp_reps=2 2>/dev/null | cut -c1-230
SX_SPX_TRACE=1 timeout -k 10 600 python tools/lp_e2e.py n1 m=1000000 n=10000000 > gpurun_out/r20_c5.json 2> gpurun_out/r20_c5_trace.txt; cut -c1-230 gpurun_out/r20_c5.json
grep -E "matching and band|band LU done|Schur complement factored|done:|sx_window" gpurun_out/r20_c5_trace.txt | cut -c1-260
