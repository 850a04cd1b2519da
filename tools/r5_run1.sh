mkdir -p gpurun_out/r5
(timeout 300 tools/block_wave_bench poisson 256 > gpurun_out/r5/relay_L0.log 2>&1; echo rc=$? >> gpurun_out/r5/relay_L0.log)
(timeout 600 python tools/block_wave_levels.py 256 1 512 > gpurun_out/r5/relay_L1.log 2>&1; echo rc=$? >> gpurun_out/r5/relay_L1.log)
(timeout 600 python bench.py > gpurun_out/r5/bench0.json 2> gpurun_out/r5/bench0.err; echo rc=$? >> gpurun_out/r5/bench0.err)
grep -E "relay|dataflow f|dataflow b|rc=" gpurun_out/r5/relay_L0.log | cut -c1-250 | head -60
grep -E "relay|dataflow f|dataflow b|rc=" gpurun_out/r5/relay_L1.log | cut -c1-250 | head -60
tail -c 1500 gpurun_out/r5/bench0.err
