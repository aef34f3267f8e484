python -m pytest tests -m gpu -x -q > gpurun_out/r27_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r27_pytest.log
tail -4 gpurun_out/r27_pytest.log; grep -n "^E " gpurun_out/r27_pytest.log | head -5
