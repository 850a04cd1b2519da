"""Blocks of right-hand sides on the dataflow layout: smoother time of the block-ordered levels per cap on the columns one
workgroup carries (tunable gs_bw_nc), and the whole V-cycle.   usage: python tools/multirhs_flow.py [N=256] [bs=8] [caps ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import amg_amd as AMG

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
caps = [int(a) for a in sys.argv[3:]] or [1, 2, 3, 4]
A = AMG.poisson((N, N, N))
ml = AMG.ruge_stuben(A, setup="gpu", device=0)
n = A.m
rng = np.random.default_rng(0)
dev = ml.device(0, bs)
lib = dev.lib
bd = AMG.DeviceBuffer(n * bs, 0, rng.random(n * bs))
zd = AMG.DeviceBuffer(n * bs, 0)
for cap in caps:
    assert lib.amgh_debug_set_tunable(b"gs_bw_nc", cap) == 0
    for _ in range(2):
        assert lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0) == 0
    lib.amgh_dev_sync(0)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    ms = 1e3 * (time.perf_counter() - t0) / reps
    dev.profile(True)
    for _ in range(2):
        lib.amgh_precond_apply_d(dev.h, bd.ptr, zd.ptr, 0)
    lib.amgh_dev_sync(0)
    prof = dev.profile_read()
    dev.profile(False)
    pre = [prof[k] for k in prof if k.lower().startswith("pre")][0]
    post = [prof[k] for k in prof if k.lower().startswith("post")][0]
    print(f"bs={bs} columns per workgroup <= {cap}: V-cycle {ms:7.2f} ms; smoothers (pre + post) level 0 {(pre[0] + post[0]) / 2:6.2f} ms, "
          f"level 1 {(pre[1] + post[1]) / 2:6.2f} ms, level 2 {(pre[2] + post[2]) / 2:6.2f} ms", flush=True)
lib.amgh_debug_set_tunable(b"gs_bw_nc", 0)
