SX_SPX_TRACE=1 timeout -k 10 900 python tools/lp_e2e.py n1 m=1000000 n=10000000 > gpurun_out/r3_c5.json 2> gpurun_out/r3_c5_trace.txt; echo "c5 rc=$?"
cat gpurun_out/r3_c5.json
grep -v "round [0-9]*: status 0" gpurun_out/r3_c5_trace.txt | tail -40
