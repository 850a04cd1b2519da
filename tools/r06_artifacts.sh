# the round-6 measurement artifacts in one GPU call: bench line (with PMC traffic, read ceiling, clock state), rocprofv3 kernel stats
# of the same command, per-level profile
mkdir -p gpurun_out/r6/final
R=$PWD
O=$R/gpurun_out/r6/final
timeout 700 python bench.py > $O/bench_256.json 2> $O/bench_256.err
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_r06 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_r06 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --light > $O/rocprof_run.log 2>&1; f=$(find /tmp/rp_r06 -name "*kernel_trace.csv" | head -1); python $R/tools/rocprof_summary.py $f $O/rocprofv3_kernel_stats_bench256.txt > /dev/null 2>&1)
timeout 600 python tools/vcycle_profile.py 256 3 > $O/vcycle_profile.log 2>&1
ls -la $O
python -c "
import json; d=json.load(open('$O/bench_256.json')); print(d['ms_per_step'], d['value'], d['setup_s'], d['roofline']['frac'], d['roofline']['frac_of_read_ceiling'], d['roofline']['traffic'], d['parity'], d['block_of_right_hand_sides'].get('ms_per_cycle'))"
head -30 $O/rocprofv3_kernel_stats_bench256.txt | cut -c1-200
# the bs = 8 cycle by columns per workgroup / record layout, with its per-level profile
timeout 900 python tools/bs_sweep.py 256 8 > $O/bs_sweep.log 2>&1; tail -12 $O/bs_sweep.log
# what a general operator gets on this build: variable coefficients, smoothed aggregation, a 2-D grid
timeout 900 python tools/other_problems.py > $O/other_problems.log 2>&1; tail -12 $O/other_problems.log
