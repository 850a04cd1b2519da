"""Setup-phase timing at 256^3 for a list of OpenMP thread counts (AMGS_TIMING=1 prints the per-level labels)."""
import sys, time
import amg_amd as AMG
from amg_amd._libs import setup_lib

L = setup_lib()
A = AMG.poisson((256, 256, 256))
for nt in [int(a) for a in sys.argv[1:]] or [0]:
    if nt:
        L.amgs_set_threads(nt)
    t = time.time()
    ml = AMG.ruge_stuben(A)
    print("threads", nt, "ruge_stuben", round(time.time() - t, 2), "s", flush=True)
    del ml
