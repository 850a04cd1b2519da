"""V-cycle time of the 256^3 hierarchy by the persistent grid of the relayed sweeps (run-time tunables gs_bw_grid / gs_bw_grid_long).
usage: python tools/grid_sweep.py [N=256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amg_amd as AMG
from bench import uniform
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
AMG.hip_lib().amgh_debug_set_tunable(b"gs_lean", 0)   # (the full footprint: both record layouts stay on the device, gs_bw_dict switches between them at run time)
ml = AMG.ruge_stuben(AMG.poisson((N, N, N)), setup="gpu", device=0)
dev = ml.device()
lib = dev.lib
n = ml.levels[0].A.m
bd, zd = AMG.DeviceBuffer(n, 0, uniform(n, 0)), AMG.DeviceBuffer(n, 0)
def cyc(reps=10):
    for _ in range(2): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    t0 = time.perf_counter()
    for _ in range(reps): lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    assert lib.amgh_dev_sync(0) == 0
    return 1e3 * (time.perf_counter() - t0) / reps
z0 = None
DEF = {b"gs_bw_grid_long": 512, b"gs_bw_grid": 0, b"gs_bw_dict": 1}
for name, vals in ((b"gs_bw_dict", (1, 0, 1)), (b"gs_bw_grid_long", (0, 256, 384, 512, 640, 768, 1024)), (b"gs_bw_grid", (0, 768, 1024, 1280, 1536))):
    for v in vals:
        lib.amgh_debug_set_tunable(name, v)
        t = cyc()
        z = zd.download()
        if z0 is None: z0 = z
        print(f"{name.decode()} = {v:5d}: {t:7.3f} ms per V-cycle, L0 / L1 presmooth {dev.bench_op(0, 4, 3, 1):.3f} / {dev.bench_op(1, 4, 3, 1):.3f} ms, bitwise the first {bool((z == z0).all())}", flush=True)
    lib.amgh_debug_set_tunable(name, DEF[name])
