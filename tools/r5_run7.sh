mkdir -p gpurun_out/r5
(BW_RELAY_ONLY=3 timeout 300 tools/block_wave_bench poisson 256 > gpurun_out/r5/relay2_L0.log 2>&1; echo rc=$? >> gpurun_out/r5/relay2_L0.log)
grep -E "^relay W|^dataflow f|^dataflow b|== relay|rc=" gpurun_out/r5/relay2_L0.log | cut -c1-200
(timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r5/pytest_gpu.log 2>&1; echo rc=$? >> gpurun_out/r5/pytest_gpu.log)
tail -15 gpurun_out/r5/pytest_gpu.log
(timeout 600 python bench.py > gpurun_out/r5/bench1.json 2> gpurun_out/r5/bench1.err; echo rc=$? >> gpurun_out/r5/bench1.err)
tail -3 gpurun_out/r5/bench1.err; python -c "
import json; d=json.load(open('gpurun_out/r5/bench1.json')); print(d['ms_per_step'], d['sweep_roofline']['cycle_ms_by_label'], d['roofline']['frac'], d['roofline']['frac_of_read_ceiling'], d['block_of_right_hand_sides'].get('ms_per_cycle'))"
