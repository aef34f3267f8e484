timeout -k 10 300 python tools/lp_e2e.py c2 2>/dev/null | cut -c1-260
timeout -k 10 120 python tools/lp_e2e.py n1 gpp_reps=2 2>/dev/null | cut -c1-200
python -m pytest tests -m gpu -x -q > gpurun_out/r25_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r25_pytest.log
tail -4 gpurun_out/r25_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
