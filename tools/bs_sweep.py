"""bs = 8 V-cycle of the 256^3 hierarchy by the columns a workgroup of the multi-column dataflow sweep carries (gs_bw_nc) and by
the record layout (gs_bw_dict), with the per-level profile of the block's cycle.   usage: python tools/bs_sweep.py [N=256] [bs=8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
from bench import uniform
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
AMG.hip_lib().amgh_debug_set_tunable(b"gs_lean", 0)   # (the full footprint: both record layouts stay on the device, gs_bw_dict switches between them at run time)
ml = AMG.ruge_stuben(AMG.poisson((N, N, N)), setup="gpu", device=0)
devb = ml.device(0, bs)
lib = devb.lib
n = ml.levels[0].A.m
Bh = np.stack([uniform(n, 100 + c) for c in range(bs)], axis=1)
Bd = AMG.DeviceBuffer(n * bs, 0, np.asfortranarray(Bh).ravel(order="F"))
Zd = AMG.DeviceBuffer(n * bs, 0)
def cyc(reps=5):
    for _ in range(2): assert lib.amgh_precond_apply_d(devb.h, Bd.ptr, Zd.ptr, 0) == 0
    lib.amgh_dev_sync(0)
    t0 = time.perf_counter()
    for _ in range(reps): lib.amgh_precond_apply_d(devb.h, Bd.ptr, Zd.ptr, 0)
    assert lib.amgh_dev_sync(0) == 0
    return 1e3 * (time.perf_counter() - t0) / reps
z0 = None
for dict_on in (1, 0):
    for nc in (2, 3, 4, 1):
        lib.amgh_debug_set_tunable(b"gs_bw_dict", dict_on); lib.amgh_debug_set_tunable(b"gs_bw_nc", nc)
        t = cyc()
        z = Zd.download()
        if z0 is None: z0 = z
        print(f"gs_bw_dict = {dict_on} gs_bw_nc = {nc}: {t:7.3f} ms per bs = {bs} cycle; L0 / L1 presmooth {devb.bench_op(0, 4, 3, 1):.3f} / {devb.bench_op(1, 4, 3, 1):.3f} ms; bitwise the first {bool((z == z0).all())}", flush=True)
lib.amgh_debug_set_tunable(b"gs_bw_dict", 1); lib.amgh_debug_set_tunable(b"gs_bw_nc", 2)
devb.profile(True)
for _ in range(3): lib.amgh_precond_apply_d(devb.h, Bd.ptr, Zd.ptr, 0)
lib.amgh_dev_sync(0)
prof = devb.profile_read()
devb.profile(False)
labs = list(prof.keys())
print("lvl " + " ".join(f"{l[:14]:>15s}" for l in labs))
for li in range(len(ml.levels) + 1):
    print(f"{li:3d} " + " ".join(f"{prof[l][li] / 3:12.3f} ms" for l in labs))
