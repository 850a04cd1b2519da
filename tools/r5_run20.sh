mkdir -p gpurun_out/r5
(timeout 900 python -m pytest tests/test_gpu_flow.py -x -q -m gpu -k "irregular" > gpurun_out/r5/pytest_irreg.log 2>&1; echo rc=$? >> gpurun_out/r5/pytest_irreg.log)
tail -15 gpurun_out/r5/pytest_irreg.log | cut -c1-300
