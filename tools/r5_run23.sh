(timeout 600 python -m pytest tests/test_gpu_abi_errors.py tests/test_gpu_abi_surface.py -x -q -m gpu 2>&1 | tail -5)
