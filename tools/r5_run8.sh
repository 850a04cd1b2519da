mkdir -p gpurun_out/r5
cat > /tmp/dump1.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import amg_amd as AMG
ml = AMG.ruge_stuben(AMG.poisson((256, 256, 256)), setup="gpu")
for li in (1,):
    M = ml.levels[li].A
    with open(f"/tmp/bw_L{li}.bin", "wb") as f:
        np.array([M.m, M.nnz], dtype=np.int64).tofile(f)
        np.asarray(M.colptr, dtype=np.int32).tofile(f)
        np.asarray(M.rowval, dtype=np.int32).tofile(f)
        np.asarray(M.nzval, dtype=np.float64).tofile(f)
PY
python /tmp/dump1.py
for g in 0 512 768 1024 1280; do
  (BW_RELAY_ONLY=3 BW_RELAY_GRID=$g timeout 300 tools/block_wave_bench poisson 256 > gpurun_out/r5/pers${g}_L0.log 2>&1; echo rc=$? >> gpurun_out/r5/pers${g}_L0.log)
  (BW_RELAY_ONLY=3 BW_RELAY_GRID=$g timeout 300 tools/block_wave_bench file /tmp/bw_L1.bin 512 > gpurun_out/r5/pers${g}_L1.log 2>&1; echo rc=$? >> gpurun_out/r5/pers${g}_L1.log)
done
for g in 0 512 768 1024 1280; do for l in L0 L1; do echo "== grid $g $l"; grep -E "^relay|== relay|20 alt|rc=" gpurun_out/r5/pers${g}_$l.log | grep -v stamps | cut -c1-220; done; done
