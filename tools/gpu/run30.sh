python -m pytest tests/test_gpu_crossover_band.py -x -q -k "config5 or headline_size" 2>&1 | tail -3
