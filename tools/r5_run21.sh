mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests/test_gpu_flow.py tests/test_gpu_float32.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r5/bench_dict2.json 2> gpurun_out/r5/bench_dict2.err; tail -3 gpurun_out/r5/bench_dict2.err
python -c "
import json; d=json.load(open('gpurun_out/r5/bench_dict2.json')); print(d['ms_per_step'], d['setup_s'], d['parity'], d['block_of_right_hand_sides'])"
