mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests/test_gpu_flow.py -x -q -m gpu 2>&1 | tail -8
timeout 1200 python tools/bs_sweep.py 256 8 2>&1 | tee gpurun_out/r5/bs_sweep_il.log | tail -30
