mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_gpu_coded.py tests/test_gpu_parity.py tests/test_gpu_float32.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r5/bench_code2.json 2> gpurun_out/r5/bench_code2.err; tail -3 gpurun_out/r5/bench_code2.err
python -c "
import json; d=json.load(open('gpurun_out/r5/bench_code2.json')); print(d['ms_per_step'], d['setup_s'], d['parity'], d['block_of_right_hand_sides']['ms_per_cycle'], d['block_of_right_hand_sides']['first_column_rel_diff_vs_single_column_cycle'], d['hbm_bytes'])"
