"""Per-level / per-label V-cycle breakdown for workspace block sizes 1 and bs (hipEvent labels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sizes = [int(a) for a in sys.argv[2:]] or [1, 8]
A = AMG.poisson((N, N, N))
ml = AMG.ruge_stuben(A)
n = A.m
rng = np.random.default_rng(0)
L = len(ml.levels)
for bs in sizes:
    dev = ml.device(0, bs)
    lib = dev.lib
    bd = AMG.DeviceBuffer(n * bs, 0, rng.random(n * bs))
    zd = AMG.DeviceBuffer(n * bs, 0)
    for _ in range(2):
        lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    dev.profile(True)
    cycles = 2
    for _ in range(cycles):
        lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    prof = dev.profile_read()
    dev.profile(False)
    print(f"bs={bs}")
    print("%-3s %10s | %s" % ("lvl", "rows", "  ".join("%-13s" % k[:13] for k in prof)))
    for l in range(L + 1):
        rows = ml.levels[l].A.m if l < L else ml.final_A.m
        print("%-3d %10d | %s" % (l, rows, "  ".join("%10.3f ms" % (prof[k][l] / cycles) for k in prof)))
    del bd, zd
