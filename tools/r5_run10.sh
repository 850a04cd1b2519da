mkdir -p gpurun_out/r5
(timeout 1200 python -m pytest tests/test_gpu_ipc.py -x -q -m gpu -k "pipelined or exact" > gpurun_out/r5/pipe_ipc_test.log 2>&1; echo rc=$? >> gpurun_out/r5/pipe_ipc_test.log)
tail -30 gpurun_out/r5/pipe_ipc_test.log | cut -c1-300
