SX_SPX_TRACE=1 timeout -k 10 120 python tools/lp_e2e.py n1 gpp_reps=2 > gpurun_out/r26_n1.json 2> gpurun_out/r26_n1_trace.txt; cut -c1-700 gpurun_out/r26_n1.json
grep -v "^\[sx_crossover_band\]   round" gpurun_out/r26_n1_trace.txt | cut -c1-220 | tail -40
