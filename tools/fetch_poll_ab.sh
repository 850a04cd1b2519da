#!/bin/bash
# The fetcher's polls pipelined (csrc/hip/gs_relay.hpp, BW_RELAY_FETCH_PD polls in flight, BW_RELAY_FETCH_GAP x 64 cycles apart) against the
# round-5 loop, on levels 0 / 1 of the 256^3 hierarchy: tools/relay_bench_<variant> built with -DBW_RELAY_FETCH_PD=.. -DBW_RELAY_FETCH_GAP=..
OUT=${1:-gpurun_out/fetch_poll_ab.log}
cat > /tmp/dump01.py <<'PY'
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
python3 /tmp/dump01.py
export BW_RELAY_ONLY=3 BW_RELAY_DICT=1 BW_RELAY_LATE=1
: > $OUT
for rep in 1 2; do
for v in ${VARIANTS:-base p2g6 p2g12 p3g4 p3g8 p4g4}; do
  for L in 0 1; do
    echo "==== $v level $L (rep $rep)" >> $OUT
    (timeout 200 tools/relay_bench_$v /tmp/bw_L$L.bin 512 1024 2>&1; echo rc=$?) | grep -E "relay W = 3|rc=" | cut -c1-200 >> $OUT
  done
done
done
