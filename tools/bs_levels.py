"""Blocks of right-hand sides on the merged-group levels: the bs-column V-cycle of the N^3 hierarchy per level (smoothers)
for a list of tunable settings, next to the single-column cycle.
usage: python tools/bs_levels.py [N=256] [bs=8] [name=value[,name=value...] ...]   (each argument one setting; "-" = defaults)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG
from bench import uniform
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
settings = sys.argv[3:] or ["-"]
ml = AMG.ruge_stuben(AMG.poisson((N, N, N)), setup="gpu", device=0)
n = ml.levels[0].A.m
nl = len(ml.levels) + 1


def run(dev, nb, tag):
    lib = dev.lib
    Bh = np.stack([uniform(n, 100 + c) for c in range(nb)], axis=1)
    Bd = AMG.DeviceBuffer(n * nb, 0, np.asfortranarray(Bh).ravel(order="F"))
    Zd = AMG.DeviceBuffer(n * nb, 0)
    for _ in range(2): assert lib.amgh_precond_apply_d(dev.h, Bd.ptr, Zd.ptr, 0) == 0
    lib.amgh_dev_sync(0)
    t0 = time.perf_counter()
    for _ in range(5): lib.amgh_precond_apply_d(dev.h, Bd.ptr, Zd.ptr, 0)
    assert lib.amgh_dev_sync(0) == 0
    t = 1e3 * (time.perf_counter() - t0) / 5
    dev.profile(True)
    for _ in range(3): lib.amgh_precond_apply_d(dev.h, Bd.ptr, Zd.ptr, 0)
    lib.amgh_dev_sync(0)
    prof = dev.profile_read()
    dev.profile(False)
    sm = [(prof["Presmoother"][l] + prof["Postsmoother"][l]) / 3 for l in range(nl)]
    print(f"{tag:40s} bs = {nb}: {t:7.3f} ms per cycle; smoothers by level: " + " ".join(f"{x:6.3f}" for x in sm[:7]), flush=True)
    return Zd.download().reshape((nb, n))


dev1 = ml.device(0, 1)
z1 = run(dev1, 1, "single column")
del dev1
devb = ml.device(0, bs)
zref = None
for st in settings:
    pairs = [] if st == "-" else [p.split("=") for p in st.split(",")]
    for k, v in pairs: assert devb.lib.amgh_debug_set_tunable(k.encode(), int(v)) == 0, k
    z = run(devb, bs, st)
    if zref is None: zref = z
    print(f"    max rel. difference from the first setting {np.abs(z - zref).max() / np.abs(zref).max():.2e}; bitwise {bool((z == zref).all())}", flush=True)
    for k, v in pairs: devb.lib.amgh_debug_set_tunable(k.encode(), {"gs_il": 1, "gs_tri_rb": 1}.get(k, 0))
