#!/usr/bin/env python3
"""tools/block_wave_bench on the levels of the ruge_stuben hierarchy of poisson(N^3): dumps each level's operator to /tmp
and runs the bench on it.   usage: python tools/block_wave_levels.py [N=256] [levels=0,1,2] [target_rows...]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
levels = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2]
targets = sys.argv[3:] or ["512"]
ml = AMG.ruge_stuben(AMG.poisson((N, N, N)), setup="gpu")
here = os.path.dirname(os.path.abspath(__file__))
for li in levels:
    M = ml.levels[li].A
    path = f"/tmp/bw_L{li}.bin"
    with open(path, "wb") as f:
        np.array([M.m, M.nnz], dtype=np.int64).tofile(f)
        np.asarray(M.colptr, dtype=np.int32).tofile(f)
        np.asarray(M.rowval, dtype=np.int32).tofile(f)
        np.asarray(M.nzval, dtype=np.float64).tofile(f)
    for t in targets:
        print(f"==== level {li}: {M.m} rows, {M.nnz} entries, target rows {t}", flush=True)
        subprocess.run([os.path.join(here, os.environ.get("BW_BENCH", "block_wave_bench")), "file", path, t], check=False)
    os.remove(path)
