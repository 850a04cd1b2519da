"""Per-level sweep time and schedule statistics of a smoothed-aggregation hierarchy, default and forced merge depths."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG
from amg_amd.device import DeviceHierarchy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 160
forces = [int(x) for x in sys.argv[2:]] or [0]
A = AMG.poisson((N, N, N)); ml = AMG.smoothed_aggregation(A); lib = AMG.hip_lib()
lv = list(range(len(ml.levels)))
lib.amgh_debug_set_tunable(b"gs_bw", 0)
for m in forces:
    lib.amgh_debug_set_tunable(b"gs_merge_force", m)
    dev = DeviceHierarchy(ml, 0, 1)
    ts = [dev.bench_op(l, 4, 3, 1) for l in lv]
    st = [dev.gs_sweep_stats(l, False) for l in lv]
    print(f"force m={m}: " + "  ".join(f"L{l} {t:7.3f}" for l, t in zip(lv, ts)),
          [(s["launches"], s["levels_per_group"], round(s["slot_entries"] / max(1, ml.levels[l].A.nnz), 2)) for l, s in zip(lv, st)], flush=True)
    del dev; gc.collect()
