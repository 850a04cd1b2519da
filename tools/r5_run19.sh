mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_gpu_flow.py tests/test_gpu_float32.py -x -q -m gpu 2>&1 | tail -8
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r5/bench_dict.json 2> gpurun_out/r5/bench_dict.err; tail -3 gpurun_out/r5/bench_dict.err; cat gpurun_out/r5/bench_dict.json | cut -c1-1500
