"""bs = 8 cycle of the 256^3 hierarchy with restriction / prolongation through the interleaved copy (tunable rhs_il = 1; per operator the
kernel amgh_finalize timed as faster) against column by column everywhere (rhs_il = 0), with the per-level times.  AMGH_VERBOSE=1 prints
the timings behind the choice.   usage: python tools/rhs_interleave_ab.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import amg_amd as AMG
from bench import uniform
N, bs = 256, 8
ml = AMG.ruge_stuben(AMG.poisson((N, N, N)), setup="gpu", device=0)
devb = ml.device(0, bs)
lib = devb.lib
n = ml.levels[0].A.m
Bh = np.stack([uniform(n, 100 + c) for c in range(bs)], axis=1)
Bd = AMG.DeviceBuffer(n * bs, 0, np.asfortranarray(Bh).ravel(order="F")); Zd = AMG.DeviceBuffer(n * bs, 0)
for il in (1, 0, 1):
    lib.amgh_debug_set_tunable(b"rhs_il", il)
    for _ in range(2): assert lib.amgh_precond_apply_d(devb.h, Bd.ptr, Zd.ptr, 0) == 0
    lib.amgh_dev_sync(0)
    t0 = time.perf_counter()
    for _ in range(5): lib.amgh_precond_apply_d(devb.h, Bd.ptr, Zd.ptr, 0)
    lib.amgh_dev_sync(0)
    t = 1e3 * (time.perf_counter() - t0) / 5
    devb.profile(True)
    for _ in range(3): lib.amgh_precond_apply_d(devb.h, Bd.ptr, Zd.ptr, 0)
    lib.amgh_dev_sync(0)
    prof = devb.profile_read(); devb.profile(False)
    print(f"rhs_il = {il}: {t:.3f} ms per bs = 8 cycle; level 0 / 1 residual {prof['Residual eval'][0] / 3:.3f} / {prof['Residual eval'][1] / 3:.3f}, restriction {prof['Restriction'][0] / 3:.3f} / {prof['Restriction'][1] / 3:.3f}, prolongation {prof['Prolongation'][0] / 3:.3f} / {prof['Prolongation'][1] / 3:.3f} ms", flush=True)
