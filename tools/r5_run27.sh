mkdir -p gpurun_out/r5
timeout 2800 python -m pytest tests -q -m gpu -rs 2>&1 | tail -12
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r5/bench_onlydict.json 2> gpurun_out/r5/bench_onlydict.err; tail -3 gpurun_out/r5/bench_onlydict.err
python -c "
import json; d=json.load(open('gpurun_out/r5/bench_onlydict.json')); print(d['ms_per_step'], d['setup_s'], d['parity'], d['block_of_right_hand_sides']['ms_per_cycle'], d['block_of_right_hand_sides']['hbm_bytes'], d['hbm_bytes'], d['hbm_bytes_by_category'])"
