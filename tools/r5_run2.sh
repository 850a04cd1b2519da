mkdir -p gpurun_out/r5
(timeout 300 tools/block_wave_bench_stamps poisson 256 > gpurun_out/r5/stamps_L0.log 2>&1; echo rc=$? >> gpurun_out/r5/stamps_L0.log)
(timeout 300 tools/block_wave_bench_d4 poisson 256 > gpurun_out/r5/d4_L0.log 2>&1; echo rc=$? >> gpurun_out/r5/d4_L0.log)
(BW_BENCH=block_wave_bench_stamps timeout 600 python tools/block_wave_levels.py 256 1 512 > gpurun_out/r5/stamps_L1.log 2>&1; echo rc=$? >> gpurun_out/r5/stamps_L1.log)
grep -E "relay W|all steps|rc=" gpurun_out/r5/stamps_L0.log | cut -c1-400
grep -E "relay W|rc=" gpurun_out/r5/d4_L0.log | cut -c1-250
grep -E "relay W|all steps|rc=" gpurun_out/r5/stamps_L1.log | cut -c1-400
