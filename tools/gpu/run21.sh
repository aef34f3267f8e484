for i in 1 2; do
SX_SPX_TRACE=1 timeout -k 10 600 python tools/lp_e2e.py n1 m=1000000 n=10000000 > gpurun_out/r21_c5.json 2> gpurun_out/r21_c5_trace.txt; cut -c1-200 gpurun_out/r21_c5.json
grep -E "host copies|columns matched|matching and band|band LU done|done:" gpurun_out/r21_c5_trace.txt | cut -c1-200
done
nproc; uptime
