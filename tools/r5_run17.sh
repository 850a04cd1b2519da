mkdir -p gpurun_out/r5
cat > /tmp/dump1.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import amg_amd as AMG
ml = AMG.ruge_stuben(AMG.poisson((256, 256, 256)), setup="gpu")
for li in (0, 1):
    M = ml.levels[li].A
    with open(f"/tmp/bw_L{li}.bin", "wb") as f:
        np.array([M.m, M.nnz], dtype=np.int64).tofile(f)
        np.asarray(M.colptr, dtype=np.int32).tofile(f); np.asarray(M.rowval, dtype=np.int32).tofile(f); np.asarray(M.nzval, dtype=np.float64).tofile(f)
PY
python /tmp/dump1.py
for t in 512 1000 1331 1728; do
  (timeout 300 tools/relay_bench /tmp/bw_L0.bin $t 2047 > gpurun_out/r5/big_L0_t$t.log 2>&1; echo rc=$? >> gpurun_out/r5/big_L0_t$t.log)
  echo "== L0 target $t"; grep -E "^n =|relay W|rc=" gpurun_out/r5/big_L0_t$t.log | cut -c1-260
done
for t in 512 1000 1500 1900; do
  (timeout 300 tools/relay_bench /tmp/bw_L1.bin $t 2047 > gpurun_out/r5/big_L1_t$t.log 2>&1; echo rc=$? >> gpurun_out/r5/big_L1_t$t.log)
  echo "== L1 target $t"; grep -E "^n =|relay W|rc=" gpurun_out/r5/big_L1_t$t.log | cut -c1-260
done
