# the round-4 measurement artifacts in one GPU call: bench line (with PMC traffic), rocprofv3 kernel stats of the same command,
# per-level profile, dataflow sweep stamps of the two finest levels, PMC traffic of the sweeps, setup timeline
mkdir -p gpurun_out/r4/final
R=$PWD
O=$R/gpurun_out/r4/final
timeout 600 python bench.py > $O/bench_256.json 2> $O/bench_256.err
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_r04 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_r04 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --light > $O/rocprof_run.log 2>&1; f=$(find /tmp/rp_r04 -name "*kernel_trace.csv" | head -1); python $R/tools/rocprof_summary.py $f $O/rocprofv3_kernel_stats_bench256.txt > /dev/null 2>&1)
timeout 600 python tools/vcycle_profile.py 256 3 > $O/vcycle_profile.log 2>&1
timeout 900 python tools/block_wave_levels.py 256 0,1 512 > $O/flow_levels.log 2>&1
(cd /tmp && export TMPDIR=/tmp && PMC="FETCH_SIZE WRITE_SIZE" timeout 900 python $R/tools/pmc_flow.py poisson 256 > $O/pmc_flow.log 2>&1)
timeout 600 python tools/verbose_build.py 256 gpu 1 > $O/setup_timeline.log 2>&1
ls -la $O
