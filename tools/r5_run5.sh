mkdir -p gpurun_out/r5
cat > /tmp/dump1.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import amg_amd as AMG
ml = AMG.ruge_stuben(AMG.poisson((256, 256, 256)), setup="gpu")
for li in (2,):
    M = ml.levels[li].A
    with open(f"/tmp/bw_L{li}.bin", "wb") as f:
        np.array([M.m, M.nnz], dtype=np.int64).tofile(f)
        np.asarray(M.colptr, dtype=np.int32).tofile(f)
        np.asarray(M.rowval, dtype=np.int32).tofile(f)
        np.asarray(M.nzval, dtype=np.float64).tofile(f)
PY
python /tmp/dump1.py
for t in 256 512; do
  (timeout 300 tools/relay_bench36 /tmp/bw_L2.bin $t > gpurun_out/r5/relay36b_L2_t$t.log 2>&1; echo rc=$? >> gpurun_out/r5/relay36b_L2_t$t.log)
done
for t in 256 512; do echo "== L2 target $t"; cat gpurun_out/r5/relay36b_L2_t$t.log | cut -c1-300; done
