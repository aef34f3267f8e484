python -m pytest tests -m gpu -x -q > gpurun_out/r10_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r10_pytest.log
grep "hostile structure" gpurun_out/r10_pytest.log; tail -4 gpurun_out/r10_pytest.log
python bench.py > gpurun_out/r10_bench.json 2> gpurun_out/r10_bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r10_bench.json
