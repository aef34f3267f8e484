SX_SPX_TRACE=1 timeout -k 10 120 python tools/lp_e2e.py n1 gpp_reps=2 > gpurun_out/r8_n1.json 2> gpurun_out/r8_n1_trace.txt; echo "n1 rc=$?"
cat gpurun_out/r8_n1.json; grep -v "^\[sx_crossover_band\]   round" gpurun_out/r8_n1_trace.txt | tail -14
for it in 2496 9984; do
SX_PDLP_ITERS=$it timeout -k 10 120 python tools/lp_e2e.py n1 gpp_reps=2 > gpurun_out/r8_n1_$it.json 2>/dev/null; cat gpurun_out/r8_n1_$it.json
done
SX_SPX_TRACE=1 timeout -k 10 600 python tools/lp_e2e.py n1 m=1000000 n=10000000 > gpurun_out/r8_c5.json 2> gpurun_out/r8_c5_trace.txt; echo "c5 rc=$?"
cat gpurun_out/r8_c5.json
grep -v "round [0-9]*: status 0" gpurun_out/r8_c5_trace.txt | tail -34
for it in 4992 9984; do
SX_PDLP_ITERS=$it SX_SPX_TRACE=1 timeout -k 10 600 python tools/lp_e2e.py n1 m=1000000 n=10000000 > gpurun_out/r8_c5_$it.json 2> gpurun_out/r8_c5_trace_$it.txt; cat gpurun_out/r8_c5_$it.json; tail -1 gpurun_out/r8_c5_trace_$it.txt
done
