mkdir -p gpurun_out/r5
(timeout 600 python tools/grid_sweep.py 256 > gpurun_out/r5/grid_sweep.log 2>&1; echo rc=$? >> gpurun_out/r5/grid_sweep.log)
grep -v "^\[amghip\]" gpurun_out/r5/grid_sweep.log | tail -16
(timeout 2700 python -m pytest tests -x -q -m gpu > gpurun_out/r5/pytest_gpu2.log 2>&1; echo rc=$? >> gpurun_out/r5/pytest_gpu2.log)
tail -8 gpurun_out/r5/pytest_gpu2.log
