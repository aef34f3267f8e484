python -m pytest tests/test_gpu_bandlu.py tests/test_gpu_denselu.py tests/test_gpu_crossover_band.py -x -q 2>&1 | tail -3
