"""Per-phase shader-clock breakdown of the block-inverse sweep (gs_block_kernel) on the levels that use it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import amg_amd as AMG

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = AMG.poisson((N, N, N))
ml = AMG.ruge_stuben(A)
dev = ml.device()
lib = dev.lib
print("levels", [l.A.m for l in ml.levels])
out = (C.c_ulonglong * 8)()
for l, pipe in [(l, p) for l in range(len(ml.levels)) for p in (0, 1)]:
    n = ml.levels[l].A.m
    if n > 100000 or n < 256:
        continue
    lib.amgh_debug_set_tunable(b"gs_block_pipe", pipe)
    dev.bench_op(l, 4, 2, 1)
    lib.amgh_debug_chain_timing(1, None)
    lib.amgh_debug_chain_timing(1, out)   # reset
    ms = dev.bench_op(l, 4, 1, 0)
    lib.amgh_debug_chain_timing(1, out)
    v = list(out)
    lib.amgh_debug_chain_timing(0, None)
    steps = max(1, v[4])
    print(f"L{l} pipe={pipe} n={n}: presmooth {ms:.3f} ms, block steps {steps}; cycles/step: load->LDS {v[0]/steps:.0f}  row sums {v[1]/steps:.0f}  "
          f"dense {v[2]/steps:.0f}  store+fence {v[3]/steps:.0f}  (sum {sum(v[:4])/steps:.0f});  row sums = loop {v[5]/steps:.0f} + shuffles {v[6]/steps:.0f} + s_vec/barrier {v[7]/steps:.0f}")
