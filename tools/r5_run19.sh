mkdir -p gpurun_out/r5
(timeout 2700 python -m pytest tests -x -q -m gpu > gpurun_out/r5/pytest_gpu3.log 2>&1; echo rc=$? >> gpurun_out/r5/pytest_gpu3.log)
tail -6 gpurun_out/r5/pytest_gpu3.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
