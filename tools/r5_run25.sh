(timeout 600 python -m pytest tests/test_gpu_float32.py -x -q -m gpu -k "pipelined" 2>&1 | tail -12 | cut -c1-300)
