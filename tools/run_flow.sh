# usage: bash tools/run_flow.sh TAG [pmc]   -> gpurun_out/r4/flow_TAG.log (+ pmc_TAG.log)
mkdir -p gpurun_out/r4
R=$PWD
timeout 120 tools/block_wave_bench poisson 256 > gpurun_out/r4/flow_$1.log 2>&1; echo rc=$? >> gpurun_out/r4/flow_$1.log
grep -A60 "== dataflow" gpurun_out/r4/flow_$1.log | cut -c1-300
if [ "$2" = "pmc" ]; then
  (cd /tmp && export TMPDIR=/tmp && PMC="FETCH_SIZE WRITE_SIZE" python $R/tools/pmc_flow.py poisson 256 > $R/gpurun_out/r4/pmc_$1.log 2>&1); cat gpurun_out/r4/pmc_$1.log
fi
