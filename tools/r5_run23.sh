mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_gpu_coded.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -12
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r5/bench_code.json 2> gpurun_out/r5/bench_code.err; tail -3 gpurun_out/r5/bench_code.err
python -c "
import json; d=json.load(open('gpurun_out/r5/bench_code.json')); print(d['ms_per_step'], d['setup_s'], d['parity'], d['block_of_right_hand_sides']['ms_per_cycle'], d['hbm_bytes'])"
timeout 600 python tools/vcycle_profile.py 256 3 2>&1 | tee gpurun_out/r5/vcycle_profile_code.log | head -24
