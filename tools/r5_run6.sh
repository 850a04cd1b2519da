mkdir -p gpurun_out/r5
(INSLOT_ONLY=1 timeout 300 tools/gs_step_bench 200 > gpurun_out/r5/inslot.log 2>&1; echo rc=$? >> gpurun_out/r5/inslot.log)
(timeout 600 tools/gs_step_bench 200 > gpurun_out/r5/gs_step_full.log 2>&1; echo rc=$? >> gpurun_out/r5/gs_step_full.log)
grep -E "^----|IN-SLOT|T512 E1 eager|T256 E2 eager|rc=" gpurun_out/r5/inslot.log | cut -c1-200
grep -E "^----|persistent|dep T512|T512 E1 eager|rc=" gpurun_out/r5/gs_step_full.log | cut -c1-200
