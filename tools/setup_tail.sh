mkdir -p gpurun_out/r4
BW_NO_FLOW=1 BW_PLAN_TIMING=1 timeout 300 tools/block_wave_bench poisson 256 2>&1 | grep "bw plan" > gpurun_out/r4/plan_laps.log
cat gpurun_out/r4/plan_laps.log
for t in 4 8 16; do
  AMGH_BW_THREADS=$t timeout 300 python bench.py --no-pmc --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('threads $t setup_s', round(d['setup_s'],2), 'ms', round(d['ms_per_step'],2))"
done
