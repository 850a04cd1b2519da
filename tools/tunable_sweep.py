"""Smoother-pass time per level for several values of a schedule-build tunable (gs_super, gs_merge, ...).
usage: python tools/tunable_sweep.py N name v1 v2 ..."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG
from amg_amd.device import DeviceHierarchy

N = int(sys.argv[1])
name = sys.argv[2].encode()
A = AMG.poisson((N, N, N))
ml = AMG.ruge_stuben(A)
lib = AMG.hip_lib()
print("levels", [l.A.m for l in ml.levels])
lv = [l for l in range(len(ml.levels)) if ml.levels[l].A.m >= 256]
for v in [int(a) for a in sys.argv[3:]]:
    lib.amgh_debug_set_tunable(name, v)
    dev = DeviceHierarchy(ml, 0, 1)
    ts = [dev.bench_op(l, 4, 3, 1) for l in lv]
    print(f"{name.decode()}={v}: " + "  ".join(f"L{l} {t:7.3f}" for l, t in zip(lv, ts)) + f"   sum {sum(ts):7.3f} ms", flush=True)
    del dev
    gc.collect()
