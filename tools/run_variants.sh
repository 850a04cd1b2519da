mkdir -p gpurun_out/r4
for v in "$@"; do
  timeout 200 tools/block_wave_bench_$v poisson 256 > gpurun_out/r4/flow_256_$v.log 2>&1; echo rc=$? >> gpurun_out/r4/flow_256_$v.log
  echo "=== $v"; grep -A100 "== dataflow" gpurun_out/r4/flow_256_$v.log | grep -E "dataflow (forward|backward) +sweep|vs the scalar|alternating|stamps|depth +(0|14|28|49|70|84|93|79|65|44|23|9):" | cut -c1-200
done
