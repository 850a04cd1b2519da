mkdir -p gpurun_out/r5
cat > /tmp/dump1.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import amg_amd as AMG
ml = AMG.ruge_stuben(AMG.poisson((256, 256, 256)), setup="gpu")
M = ml.levels[1].A
with open("/tmp/bw_L1.bin", "wb") as f:
    np.array([M.m, M.nnz], dtype=np.int64).tofile(f)
    np.asarray(M.colptr, dtype=np.int32).tofile(f); np.asarray(M.rowval, dtype=np.int32).tofile(f); np.asarray(M.nzval, dtype=np.float64).tofile(f)
PY
python /tmp/dump1.py
(BW_RELAY_ONLY=3 timeout 300 tools/block_wave_bench_stamps poisson 256 > gpurun_out/r5/xcc_L0.log 2>&1)
(BW_RELAY_ONLY=3 timeout 300 tools/block_wave_bench_stamps file /tmp/bw_L1.bin 512 > gpurun_out/r5/xcc_L1.log 2>&1)
grep -E "by placement|hand-offs \(row" gpurun_out/r5/xcc_L0.log gpurun_out/r5/xcc_L1.log | cut -c1-300
