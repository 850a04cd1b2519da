mkdir -p gpurun_out/r5
(timeout 900 python tools/dist_local_bench.py 256 4 2000000 > gpurun_out/r5/dist_local4.log 2>&1; echo rc=$? >> gpurun_out/r5/dist_local4.log)
(timeout 900 python tools/dist_local_bench.py 256 2 2000000 > gpurun_out/r5/dist_local2.log 2>&1; echo rc=$? >> gpurun_out/r5/dist_local2.log)
grep -v "^\[amghip\]" gpurun_out/r5/dist_local4.log | tail -8 | cut -c1-250
grep -v "^\[amghip\]" gpurun_out/r5/dist_local2.log | tail -8 | cut -c1-250
