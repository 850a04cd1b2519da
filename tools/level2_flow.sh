#!/bin/bash
# Level 2 of the 256^3 hierarchy (1.4 M rows of <= 35 entries, 870 dependency levels) as a dataflow of blocks under 36-entry records:
# the relayed walk with the dictionary layout and the dependency-aware row sum (round 6), offset classes by the plan or by hand.
# Needs tools/relay_bench36 (hipcc ... -DBW_PLAN_MAXK=36 -DBW_EXTRA_MAXK=36 -DBW_EXTRA_DICT tools/relay_bench.hip).  Output: $1 (a log file).
OUT=${1:-gpurun_out/level2_flow.log}
cat > /tmp/dump2.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import amg_amd as AMG
ml = AMG.ruge_stuben(AMG.poisson((256, 256, 256)), setup="gpu")
M = ml.levels[2].A
with open("/tmp/bw_L2.bin", "wb") as f:
    np.array([M.m, M.nnz], dtype=np.int64).tofile(f)
    np.asarray(M.colptr, dtype=np.int32).tofile(f); np.asarray(M.rowval, dtype=np.int32).tofile(f); np.asarray(M.nzval, dtype=np.float64).tofile(f)
PY
python3 /tmp/dump2.py
export BW_RELAY_ONLY=3
: > $OUT
for cuts in default 6,99; do
  for t in 512 768 1024; do
    for late in 0 1; do
      echo "==== cuts $cuts, target rows $t, BW_RELAY_LATE=$late" >> $OUT
      if [ $cuts = default ]; then unset BW_PLAN_CUTS; else export BW_PLAN_CUTS=$cuts; fi
      (BW_RELAY_LATE=$late timeout 200 tools/relay_bench36 /tmp/bw_L2.bin $t 2047 2>&1; echo rc=$?) | grep -E "blocks|mailboxes|dictionary|records|relay W = 3|rc=|late_ok|no d" | cut -c1-250 >> $OUT
    done
  done
done
