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
export BW_RELAY_ONLY=3
for v in D2 D3 D4; do
  for L in 0 1; do
    for g in 0 512; do
      (BW_RELAY_GRID=$g timeout 300 tools/relay_bench_$v /tmp/bw_L$L.bin 512 1024 > gpurun_out/r5/dictD_${v}_L${L}_g$g.log 2>&1; echo rc=$? >> gpurun_out/r5/dictD_${v}_L${L}_g$g.log)
      echo "== $v L$L grid $g"; grep -E "dictionary layout|relay W = 3 (f|b)|rc=|----" gpurun_out/r5/dictD_${v}_L${L}_g$g.log | cut -c1-200
    done
  done
done
timeout 900 python tools/grid_sweep.py 256 2>&1 | tee gpurun_out/r5/grid_sweep_dict.log | tail -30
