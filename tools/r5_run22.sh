mkdir -p gpurun_out/r5
(timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_ipc.py tests/test_gpu_dist.py tests/test_gpu_float32.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r5/pytest_dist.log 2>&1; echo rc=$? >> gpurun_out/r5/pytest_dist.log)
tail -5 gpurun_out/r5/pytest_dist.log | cut -c1-300
