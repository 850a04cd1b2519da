mkdir -p gpurun_out/r5
cat > /tmp/dump1.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import amg_amd as AMG
ml = AMG.ruge_stuben(AMG.poisson((256, 256, 256)), setup="gpu")
for li in (1, 2, 3):
    M = ml.levels[li].A
    with open(f"/tmp/bw_L{li}.bin", "wb") as f:
        np.array([M.m, M.nnz], dtype=np.int64).tofile(f)
        np.asarray(M.colptr, dtype=np.int32).tofile(f)
        np.asarray(M.rowval, dtype=np.int32).tofile(f)
        np.asarray(M.nzval, dtype=np.float64).tofile(f)
PY
python /tmp/dump1.py
for v in pollv2 fns nosleep spin8 all; do
  (BW_RELAY_ONLY=3 timeout 300 tools/block_wave_bench_$v poisson 256 > gpurun_out/r5/${v}_L0.log 2>&1; echo rc=$? >> gpurun_out/r5/${v}_L0.log)
  (BW_RELAY_ONLY=3 timeout 300 tools/block_wave_bench_$v file /tmp/bw_L1.bin 512 > gpurun_out/r5/${v}_L1.log 2>&1; echo rc=$? >> gpurun_out/r5/${v}_L1.log)
done
for t in 128 256 512; do
  (timeout 300 tools/relay_bench36 /tmp/bw_L2.bin $t > gpurun_out/r5/relay36_L2_t$t.log 2>&1; echo rc=$? >> gpurun_out/r5/relay36_L2_t$t.log)
done
for v in pollv2 fns nosleep spin8 all; do for l in L0 L1; do echo "== ${v}_$l"; grep -E "^relay W|^dataflow f|^dataflow b|rc=" gpurun_out/r5/${v}_$l.log | grep -v stamps | cut -c1-200; done; done
for t in 128 256 512; do echo "== L2 target $t"; cat gpurun_out/r5/relay36_L2_t$t.log | cut -c1-300; done
