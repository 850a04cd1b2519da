import gc, os, sys
sys.path.insert(0, "/root/repo")
import amg_amd as AMG
from amg_amd.device import DeviceHierarchy
N = 256
A = AMG.poisson((N, N, N)); ml = AMG.ruge_stuben(A); lib = AMG.hip_lib()
lv = [0, 1, 2, 3]
lib.amgh_debug_set_tunable(b"gs_merge_force_maxn", 9000000)
for m in (0, 2, 3, 4, 5):
    lib.amgh_debug_set_tunable(b"gs_merge_force", m)
    dev = DeviceHierarchy(ml, 0, 1)
    ts = [dev.bench_op(l, 4, 3, 1) for l in lv]
    print(f"force m={m}: " + "  ".join(f"L{l} {t:7.3f}" for l, t in zip(lv, ts)), [dev.gs_sweep_stats(l, False)["launches"] for l in lv], flush=True)
    del dev; gc.collect()
