"""Blocks per superblock in the block-inverse sweeps (read when a smoother schedule is built)."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG
from amg_amd.device import DeviceHierarchy

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = AMG.poisson((N, N, N))
ml = AMG.ruge_stuben(A)
lib = AMG.hip_lib()
print("levels", [l.A.m for l in ml.levels])
lv = [l for l in range(len(ml.levels)) if 256 <= ml.levels[l].A.m <= 100000]
for S in [int(a) for a in sys.argv[2:]] or [0, 4, 8, 16, 32]:
    lib.amgh_debug_set_tunable(b"gs_super", S)
    dev = DeviceHierarchy(ml, 0, 1)
    ts = [dev.bench_op(l, 4, 3, 1) for l in lv]
    print(f"super={S}: " + "  ".join(f"L{l} {t:7.3f}" for l, t in zip(lv, ts)) + f"   sum {sum(ts):7.3f} ms", flush=True)
    del dev
    gc.collect()
