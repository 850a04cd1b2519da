"""XCD-contiguous workgroup mapping of the stream kernel (tunable stream_xcd = 0: never; 1: where amgh_finalize found it faster — the restriction of level 1 at 256^3): SpMV / residual / restriction /
prolongation of levels 0-2 of the N^3 hierarchy, ms per launch.  python tools/stream_xcd_levels.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = AMG.hip_lib()
ml = AMG.ruge_stuben(AMG.poisson((N, N, N)), setup="gpu", device=0)
dev = ml.device()
for v in (0, 1, 0, 1):
    lib.amgh_debug_set_tunable(b"stream_xcd", v)
    for l in range(3):
        print(f"stream_xcd = {v} level {l} ({ml.levels[l].A.m} rows, {ml.levels[l].A.nnz / ml.levels[l].A.m:.1f} entries/row): "
              f"SpMV {dev.bench_op(l, 0, 20, 3):.4f}  residual {dev.bench_op(l, 3, 20, 3):.4f}  R {dev.bench_op(l, 2, 20, 3):.4f}  P {dev.bench_op(l, 1, 20, 3):.4f} ms", flush=True)
lib.amgh_debug_set_tunable(b"stream_xcd", 0)
