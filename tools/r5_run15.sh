mkdir -p gpurun_out/r5
R=$PWD
(BW_RELAY_ONLY=3 timeout 300 tools/block_wave_bench poisson 256 > gpurun_out/r5/far0_L0.log 2>&1; echo rc=$? >> gpurun_out/r5/far0_L0.log)
(BW_RELAY_ONLY=3 BW_FAR_CELLS=1 timeout 300 tools/block_wave_bench poisson 256 > gpurun_out/r5/far1_L0.log 2>&1; echo rc=$? >> gpurun_out/r5/far1_L0.log)
grep -E "alternating|rc=" gpurun_out/r5/far0_L0.log | cut -c1-200
grep -E "alternating|rc=" gpurun_out/r5/far1_L0.log | cut -c1-200
(cd /tmp && export TMPDIR=/tmp && BW_RELAY_ONLY=3 PMC="FETCH_SIZE WRITE_SIZE" timeout 600 python $R/tools/pmc_flow.py poisson 256 > $R/gpurun_out/r5/pmc_far0.log 2>&1)
(cd /tmp && export TMPDIR=/tmp && BW_RELAY_ONLY=3 BW_FAR_CELLS=1 PMC="FETCH_SIZE WRITE_SIZE" timeout 600 python $R/tools/pmc_flow.py poisson 256 > $R/gpurun_out/r5/pmc_far1.log 2>&1)
cat gpurun_out/r5/pmc_far0.log gpurun_out/r5/pmc_far1.log | cut -c1-300
(timeout 900 python -m pytest tests/test_gpu_flow.py -x -q -m gpu > gpurun_out/r5/pytest_flow.log 2>&1; echo rc=$? >> gpurun_out/r5/pytest_flow.log); tail -3 gpurun_out/r5/pytest_flow.log
(timeout 600 python tools/grid_sweep.py 256 2>&1 | grep -v "^\[amghip\]" | head -3)
