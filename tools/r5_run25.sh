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
export BW_RELAY_DICT=1
for L in 0 1; do
  for t in 256 384 512 768 1000; do
    (timeout 300 tools/relay_bench /tmp/bw_L$L.bin $t 1024 > gpurun_out/r5/dictT_L${L}_t$t.log 2>&1; echo rc=$? >> gpurun_out/r5/dictT_L${L}_t$t.log)
    echo "== L$L target $t"; grep -E "^n =|dictionary layout|relay W = . (f|b)|rc=" gpurun_out/r5/dictT_L${L}_t$t.log | cut -c1-230
  done
done
python tools/pmc_relay.py /tmp/bw_L0.bin 512 1024 2>&1 | tee gpurun_out/r5/pmc_relay_L0.log
python tools/pmc_relay.py /tmp/bw_L1.bin 512 1024 2>&1 | tee gpurun_out/r5/pmc_relay_L1.log
