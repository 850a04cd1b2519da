#!/usr/bin/env python3
"""A/B of the host C/F splitting (amgs_rs_cf_splitting_patterns) on the real strength patterns of every level of
ruge_stuben(poisson(N^3)): plain arrays vs packed records vs packed records + dry prefetch passes.  Same result
required.   usage: python tools/split_time.py [N=256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
from amg_amd._libs import setup_lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = setup_lib()
orig = L.amgs_rs_cf_splitting_patterns
VARIANTS = (("plain", {"AMGS_SPLIT_PLAIN": "1"}), ("packed+dry 4K pages", {"AMGS_NO_HUGEPAGES": "1"}), ("packed+dry huge pages", {}),
            ("+ pattern copies", {"AMGS_PATTERN_COPY": "1"}))


def timed(n, Sp, Sj, Tp, Tj, out):
    ref = None
    line = [f"n={n:9d}"]
    for name, env in VARIANTS * 2:
        for k in ("AMGS_SPLIT_PLAIN", "AMGS_SPLIT_NODRY", "AMGS_NO_HUGEPAGES", "AMGS_PATTERN_COPY"):
            os.environ.pop(k, None)
        os.environ.update(env)
        t0 = time.perf_counter()
        rc = orig(n, Sp, Sj, Tp, Tj, out)
        dt = time.perf_counter() - t0
        res = np.ctypeslib.as_array((__import__("ctypes").c_int32 * n).from_address(out)).copy()
        if ref is None:
            ref = res
        assert rc == 0 and np.array_equal(ref, res), name
        line.append(f"{name} {dt:6.3f} s")
    for k in ("AMGS_SPLIT_PLAIN", "AMGS_SPLIT_NODRY", "AMGS_NO_HUGEPAGES", "AMGS_PATTERN_COPY"):
        os.environ.pop(k, None)
    print("  ".join(line), flush=True)
    return 0


class Proxy:
    def __getattr__(self, k):
        return timed if k == "amgs_rs_cf_splitting_patterns" else getattr(L, k)


import amg_amd.hierarchy as H
H.setup_lib = lambda: Proxy()
ml = AMG.ruge_stuben(AMG.poisson((N, N, N)), setup="gpu")
print("levels", [l.A.m for l in ml.levels] + [ml.final_A.m])
