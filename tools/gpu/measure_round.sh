# The measurement pass behind profiles/rNN (run on the GPU box from the repo root: gpurun -- "mkdir -p gpurun_out && bash tools/gpu/measure_round.sh"):
# quick parity subset, the bench line, kernel traces of bench.py and of one LP crossover at both sizes, the three PMC passes.
python -m pytest tests/test_gpu_rowblock.py tests/test_gpu_lp_parity.py tests/test_gpu_property.py -x -q 2>&1 | tail -2
SX_SPX_TRACE=1 python bench.py > gpurun_out/m_bench.json 2> gpurun_out/m_bench.err; echo "bench rc=$?"
grep "sx_window" gpurun_out/m_bench.err | head
bash tools/trace_bench.sh > gpurun_out/m_trace.txt 2>&1; tail -3 gpurun_out/m_trace.txt | cut -c1-200
bash tools/prof_lp.sh n1 n1 gpp_reps=2 > gpurun_out/m_prof_n1.txt 2>&1; tail -2 gpurun_out/m_prof_n1.txt | cut -c1-300
bash tools/prof_lp.sh c5 n1 m=1000000 n=10000000 > gpurun_out/m_prof_c5.txt 2>&1; tail -2 gpurun_out/m_prof_c5.txt | cut -c1-300
bash tools/pmc_bench.sh > gpurun_out/m_pmc.txt 2>&1; tail -5 gpurun_out/m_pmc.txt | cut -c1-200
