mkdir -p gpurun_out/r5
(AMGH_VERBOSE=1 timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "pipelined" > gpurun_out/r5/pipe_test.log 2>&1; echo rc=$? >> gpurun_out/r5/pipe_test.log)
tail -40 gpurun_out/r5/pipe_test.log | cut -c1-250
