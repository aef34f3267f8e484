bash tools/gpu/prof_inbench.sh > gpurun_out/r11_prof.txt 2>&1; tail -5 gpurun_out/r11_prof.txt | cut -c1-700
for mode in band dense; do for sz in "3000 30000" "20000 200000"; do set -- $sz
SX_LP_CROSSOVER=$mode timeout -k 10 200 python tools/lp_e2e.py n1 m=$1 n=$2 gpp_reps=2 2>/dev/null | cut -c1-260
done; done
