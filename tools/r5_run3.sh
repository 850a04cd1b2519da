mkdir -p gpurun_out/r5
cat > /tmp/dump1.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import amg_amd as AMG
ml = AMG.ruge_stuben(AMG.poisson((256, 256, 256)), setup="gpu")
for li in (1, 2):
    M = ml.levels[li].A
    with open(f"/tmp/bw_L{li}.bin", "wb") as f:
        np.array([M.m, M.nnz], dtype=np.int64).tofile(f)
        np.asarray(M.colptr, dtype=np.int32).tofile(f)
        np.asarray(M.rowval, dtype=np.int32).tofile(f)
        np.asarray(M.nzval, dtype=np.float64).tofile(f)
PY
python /tmp/dump1.py
for v in stamps; do
  (timeout 300 tools/block_wave_bench_$v poisson 256 > gpurun_out/r5/${v}2_L0.log 2>&1; echo rc=$? >> gpurun_out/r5/${v}2_L0.log)
  (timeout 300 tools/block_wave_bench_$v file /tmp/bw_L1.bin 512 > gpurun_out/r5/${v}2_L1.log 2>&1; echo rc=$? >> gpurun_out/r5/${v}2_L1.log)
done
(timeout 300 tools/block_wave_bench_d2 poisson 256 > gpurun_out/r5/d2_L0.log 2>&1; echo rc=$? >> gpurun_out/r5/d2_L0.log)
(timeout 300 tools/block_wave_bench_d2 file /tmp/bw_L1.bin 512 > gpurun_out/r5/d2_L1.log 2>&1; echo rc=$? >> gpurun_out/r5/d2_L1.log)
for pad in 24 40 72; do
  (BW_RELAY_ONLY=3 BW_RELAY_LDS_PAD=$pad timeout 300 tools/block_wave_bench poisson 256 > gpurun_out/r5/pad${pad}_L0.log 2>&1; echo rc=$? >> gpurun_out/r5/pad${pad}_L0.log)
  (BW_RELAY_ONLY=3 BW_RELAY_LDS_PAD=$pad timeout 300 tools/block_wave_bench file /tmp/bw_L1.bin 512 > gpurun_out/r5/pad${pad}_L1.log 2>&1; echo rc=$? >> gpurun_out/r5/pad${pad}_L1.log)
done
for f in stamps2_L0 stamps2_L1 d2_L0 d2_L1 pad24_L0 pad24_L1 pad40_L0 pad40_L1 pad72_L0 pad72_L1; do echo "== $f"; grep -E "^relay W|hand-offs|all steps|differ|rc=" gpurun_out/r5/$f.log | grep -v "20 alt" | cut -c1-330; done
